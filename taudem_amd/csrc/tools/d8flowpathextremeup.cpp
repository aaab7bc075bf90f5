// d8flowpathextremeup -p p -sa sa -ssa ssa [-o outlets] [-lyrname n] [-lyrno i] [-min] [-nc]   (flag surface of src/D8FlowPathExtremeUpmn.cpp:50-190)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple Use:\n %s <basefilename>\n", prog);
    printf("Use with specific file names:\n %s -p <pfile> -sa <safile> -ssa <ssafile> [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] [-min] [-nc]\n", prog);
    printf("  <pfile>    D8 flow direction input\n");
    printf("  <safile>   the grid whose upstream extreme value is sought (input)\n");
    printf("  <ssafile>  the extreme value upstream of each cell (output)\n");
    printf("  -min       take the minimum (default: maximum)\n");
    printf("  -nc        do not check for edge contamination\n");
    printf("With the simple form the suffixes p, sa and ssa are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string pfile, safile, ssafile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, lyrno = 0, usemax = 1, contcheck = 1;
    if (argc < 2) usage(argv[0]);
    if (argc == 2) { pfile = cli::nameadd(argv[1], "p"); safile = cli::nameadd(argv[1], "sa"); ssafile = cli::nameadd(argv[1], "ssa"); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-p")) { if (!a.value(pfile)) usage(argv[0]); }
        else if (a.is("-sa")) { if (!a.value(safile)) usage(argv[0]); }
        else if (a.is("-ssa")) { if (!a.value(ssafile)) usage(argv[0]); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-min")) { a.flag(); usemax = 0; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    const int err = tdx_tool_d8flowpathextremeup(pfile.c_str(), safile.c_str(), ssafile.c_str(), usemax, datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno,
                                                 useOutlets, contcheck);
    return cli::finish("d8flowpathextremeup", err);
}
