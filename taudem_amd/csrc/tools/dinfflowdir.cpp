// dinfflowdir -fel f -ang a -slp s [-sfdr f]   (flag surface of src/DinfFlowDirmn.cpp:54-147)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -fel <demfile> -ang <angfile> -slp <slopefile> [-sfdr <flowfile>]\n", prog);
    printf("  <demfile>    pit-filled elevation input\n");
    printf("  <angfile>    D-infinity flow angle output (radians counter-clockwise from east)\n");
    printf("  <slopefile>  D-infinity slope output\n");
    printf("With the simple form the suffixes fel, ang and slp are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string demfile, angfile, slopefile, flowfile;
    int useflowfile = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-fel")) { if (!a.value(demfile)) usage(argv[0]); }
        else if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-slp")) { if (!a.value(slopefile)) usage(argv[0]); }
        else if (a.is("-sfdr")) { if (!a.value(flowfile)) usage(argv[0]); useflowfile = 1; }
        else usage(argv[0]);
    }
    if (argc == 2) {
        demfile = cli::nameadd(argv[1], "fel");
        angfile = cli::nameadd(argv[1], "ang");
        slopefile = cli::nameadd(argv[1], "slp");
    }
    const int err = tdx_tool_dinfflowdir(demfile.c_str(), angfile.c_str(), slopefile.c_str(), flowfile.c_str(), useflowfile);
    return cli::finish("Dinfflowdir", err);
}
