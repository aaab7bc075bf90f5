// areadinf -ang a -sca s [-o outlets] [-lyrname n] [-lyrno i] [-wg w] [-nc]   (flag surface of src/areadinfmn.cpp:49-178)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -sca <scafile> [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] [-wg <wfile>] [-nc]\n", prog);
    printf("  <angfile>     D-infinity flow angle input\n");
    printf("  <scafile>     D-infinity specific catchment area output\n");
    printf("  <outletfile>  optional outlet points; only their catchments are evaluated\n");
    printf("  <wfile>       optional weight grid\n");
    printf("  -nc           do not check for edge contamination\n");
    printf("With the simple form the suffixes ang and sca are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, scafile, wfile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, usew = 0, contcheck = 1, lyrno = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-sca")) { if (!a.value(scafile)) usage(argv[0]); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-wg")) { if (!a.value(wfile)) usage(argv[0]); usew = 1; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    if (argc == 2) { angfile = cli::nameadd(argv[1], "ang"); scafile = cli::nameadd(argv[1], "sca"); }
    const int err = tdx_tool_areadinf(angfile.c_str(), scafile.c_str(), datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno, wfile.c_str(), useOutlets, usew, contcheck);
    return cli::finish("area", err);
}
