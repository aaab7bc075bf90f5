// gridnet -p p -plen plen -tlen tlen -gord gord [-o outlets] [-lyrname n] [-lyrno i] [-mask m -thresh t]   (flag surface of src/gridnetmn.cpp:50-215)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple Usage:\n %s <basefilename>\n", prog);
    printf("Usage with specific file names:\n %s -p <pfile> -plen <plenfile> -tlen <tlenfile> -gord <gordfile> [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] "
           "[-mask <maskfile> -thresh <threshold>]\n", prog);
    printf("  <pfile>     D8 flow direction input\n");
    printf("  <plenfile>  longest flow length upstream of each cell (output)\n");
    printf("  <tlenfile>  total path length upstream of each cell (output)\n");
    printf("  <gordfile>  Strahler order of the grid network (output)\n");
    printf("  <maskfile>  optional mask grid: only cells whose mask value (read as 4-byte integer) is >= <threshold> are evaluated;\n");
    printf("              -thresh has to follow the mask file immediately\n");
    printf("With the simple form the suffixes p, plen, tlen and gord are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string pfile, plenfile, tlenfile, gordfile, maskfile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, lyrno = 0, useMask = 0, thresh = 0;
    if (argc < 2) { printf("Error: To run this program, use either the Simple Usage option or\nthe Usage with Specific file names option\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-p")) { if (!a.value(pfile)) usage(argv[0]); }
        else if (a.is("-plen")) { if (!a.value(plenfile)) usage(argv[0]); }
        else if (a.is("-tlen")) { if (!a.value(tlenfile)) usage(argv[0]); }
        else if (a.is("-gord")) { if (!a.value(gordfile)) usage(argv[0]); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-mask")) {
            if (!a.value(maskfile)) usage(argv[0]);
            useMask = 1;
            if (a.more() && a.is("-thresh")) { if (!a.value(thresh)) usage(argv[0]); }   // src/gridnetmn.cpp:157-164: -thresh must follow
            else usage(argv[0]);
        }
        else usage(argv[0]);
    }
    if (argc == 2) {
        pfile = cli::nameadd(argv[1], "p"); plenfile = cli::nameadd(argv[1], "plen");
        tlenfile = cli::nameadd(argv[1], "tlen"); gordfile = cli::nameadd(argv[1], "gord");
    }
    const int err = tdx_tool_gridnet(pfile.c_str(), plenfile.c_str(), tlenfile.c_str(), gordfile.c_str(), maskfile.c_str(), datasrc.c_str(), lyrname.c_str(),
                                     uselyrname, lyrno, useMask, useOutlets, thresh);
    return cli::finish("gridnet", err);
}
