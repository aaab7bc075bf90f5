// dinfconclimaccum -ang a -dg g -dm d -q q -ctpt c [-csol s] [-o outlets] [-lyrname n] [-lyrno i] [-nc]   (flag surface of src/DinfConcLimAccummn.cpp:49-200)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -dg <indicatorfile> -dm <dmfile> -q <qfile> -ctpt <ctptfile> [-csol <csol>] [-o <outletfile>] [-lyrname <name>] [-lyrno <n>] [-nc]\n",
           prog);
    printf("  <angfile>        D-infinity flow direction input\n");
    printf("  <indicatorfile>  disturbance indicator grid input (1 = source cell at concentration <csol>)\n");
    printf("  <dmfile>         decay multiplier grid input\n");
    printf("  <qfile>          specific discharge grid input\n");
    printf("  <ctptfile>       concentration output\n");
    printf("  <csol>           concentration of the source cells (default 1)\n");
    printf("  <outletfile>     optional outlet points; only their catchments are evaluated\n");
    printf("  -nc              do not check for edge contamination\n");
    printf("With the simple form the suffixes ang, dg, dm, q and ctpt are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, dgfile, dmfile, qfile, ctptfile, datasrc, lyrname, csol_text;
    int useOutlets = 0, uselyrname = 0, contcheck = 1, lyrno = 0;
    float csol = 1.0f;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-dg")) { if (!a.value(dgfile)) usage(argv[0]); }
        else if (a.is("-dm")) { if (!a.value(dmfile)) usage(argv[0]); }
        else if (a.is("-ctpt")) { if (!a.value(ctptfile)) usage(argv[0]); }
        else if (a.is("-q")) { if (!a.value(qfile)) usage(argv[0]); }
        else if (a.is("-csol")) { if (!a.value(csol_text)) usage(argv[0]); sscanf(csol_text.c_str(), "%f", &csol); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    if (argc == 2) {
        angfile = cli::nameadd(argv[1], "ang"); dgfile = cli::nameadd(argv[1], "dg"); dmfile = cli::nameadd(argv[1], "dm");
        qfile = cli::nameadd(argv[1], "q"); ctptfile = cli::nameadd(argv[1], "ctpt");
    }
    const int err = tdx_tool_dinfconclimaccum(angfile.c_str(), ctptfile.c_str(), dmfile.c_str(), datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno, qfile.c_str(),
                                              dgfile.c_str(), useOutlets, contcheck, csol);
    return cli::finish("area", err);
}
