// aread8 -p p -ad8 a [-o outlets] [-lyrname n] [-lyrno i] [-wg w] [-nc]   (flag surface of src/aread8mn.cpp:49-176)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -p <pfile> -ad8 <afile> [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] [-wg <wfile>] [-nc]\n", prog);
    printf("  <pfile>       D8 flow direction input\n");
    printf("  <afile>       D8 contributing area output\n");
    printf("  <outletfile>  optional outlet points (.shp, .geojson/.json or 'x y' text); only their catchments are evaluated\n");
    printf("  <wfile>       optional weight grid\n");
    printf("  -nc           do not check for edge contamination\n");
    printf("With the simple form the suffixes p and ad8 are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string pfile, afile, wfile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, usew = 0, contcheck = 1, lyrno = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-p")) { if (!a.value(pfile)) usage(argv[0]); }
        else if (a.is("-ad8")) { if (!a.value(afile)) usage(argv[0]); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-wg")) { if (!a.value(wfile)) usage(argv[0]); usew = 1; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    if (argc == 2) { afile = cli::nameadd(argv[1], "ad8"); pfile = cli::nameadd(argv[1], "p"); }
    const int err = tdx_tool_aread8(pfile.c_str(), afile.c_str(), datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno, wfile.c_str(), useOutlets, usew, contcheck);
    return cli::finish("area", err);
}
