// dinfdecayaccum -ang a -dm d -dsca o [-wg w] [-o outlets] [-lyrname n] [-lyrno i] [-nc]   (flag surface of src/DinfDecayAccummn.cpp:51-192)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -dm <dmfile> -dsca <dscafile> [-wg <wfile>] [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] [-nc]\n", prog);
    printf("  <angfile>     D-infinity flow angle input\n");
    printf("  <dmfile>      decay multiplier grid input\n");
    printf("  <dscafile>    decayed specific catchment area output\n");
    printf("  <wfile>       optional weight grid\n");
    printf("  <outletfile>  optional outlet points; only their catchments are evaluated\n");
    printf("  -nc           do not check for edge contamination\n");
    printf("With the simple form the suffixes ang, dm and dsca are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, dmfile, dscafile, wfile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, usew = 0, contcheck = 1, lyrno = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-dm")) { if (!a.value(dmfile)) usage(argv[0]); }
        else if (a.is("-dsca")) { if (!a.value(dscafile)) usage(argv[0]); }
        else if (a.is("-wg")) { if (!a.value(wfile)) usage(argv[0]); usew = 1; }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    if (argc == 2) { angfile = cli::nameadd(argv[1], "ang"); dmfile = cli::nameadd(argv[1], "dm"); dscafile = cli::nameadd(argv[1], "dsca"); }
    const int err = tdx_tool_dinfdecayaccum(angfile.c_str(), dscafile.c_str(), dmfile.c_str(), datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno,
                                            wfile.c_str(), useOutlets, usew, contcheck);
    return cli::finish("area", err);
}
