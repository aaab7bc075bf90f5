// dinftranslimaccum -ang a -tsup s -tc c -tla t -tdep d [-cs cin -ctpt cout] [-o outlets] [-lyrname n] [-lyrno i] [-nc]   (flag surface of src/DinfTransLimAccummn.cpp:49-231)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -tsup <tsupfile> -tc <tcfile> [-cs <csfile> -ctpt <ctptfile>] -tla <tlafile> -tdep <tdepfile> [-o <outletfile>] [-lyrname <name>] "
           "[-lyrno <n>] [-nc]\n", prog);
    printf("  <angfile>     D-infinity flow direction input\n");
    printf("  <tsupfile>    transport supply grid input\n");
    printf("  <tcfile>      transport capacity grid input\n");
    printf("  <csfile>      optional concentration grid input\n");
    printf("  <ctptfile>    optional concentration output (evaluated only when both <csfile> and <ctptfile> are given)\n");
    printf("  <tlafile>     transport limited accumulation output\n");
    printf("  <tdepfile>    deposition output\n");
    printf("  <outletfile>  optional outlet points; only their catchments are evaluated\n");
    printf("  -nc           do not check for edge contamination\n");
    printf("With the simple form the suffixes ang, tsup, tc, tla and tdep are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, tsupfile, tcfile, tlafile, depfile, cinfile, coutfile, datasrc, lyrname;
    int useOutlets = 0, uselyrname = 0, contcheck = 1, lyrno = 0, usec = 0, compctpt = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-tsup")) { if (!a.value(tsupfile)) usage(argv[0]); }
        else if (a.is("-tc")) { if (!a.value(tcfile)) usage(argv[0]); }
        else if (a.is("-cs")) { if (!a.value(cinfile)) usage(argv[0]); usec = 1; }
        else if (a.is("-ctpt")) { if (!a.value(coutfile)) usage(argv[0]); compctpt = 1; }
        else if (a.is("-tla")) { if (!a.value(tlafile)) usage(argv[0]); }
        else if (a.is("-tdep")) { if (!a.value(depfile)) usage(argv[0]); }
        else if (a.is("-o")) { if (!a.value(datasrc)) usage(argv[0]); useOutlets = 1; }
        else if (a.is("-lyrno")) { if (!a.value(lyrno)) usage(argv[0]); }
        else if (a.is("-lyrname")) { if (!a.value(lyrname)) usage(argv[0]); uselyrname = 1; }
        else if (a.is("-nc")) { a.flag(); contcheck = 0; }
        else usage(argv[0]);
    }
    if (argc == 2) {
        angfile = cli::nameadd(argv[1], "ang"); tsupfile = cli::nameadd(argv[1], "tsup"); tcfile = cli::nameadd(argv[1], "tc");
        tlafile = cli::nameadd(argv[1], "tla"); depfile = cli::nameadd(argv[1], "tdep");
    }
    usec = usec * compctpt;   // the concentration is evaluated only when both its input and its output are named (src/DinfTransLimAccummn.cpp:202)
    const int err = tdx_tool_dinftranslimaccum(angfile.c_str(), tsupfile.c_str(), tcfile.c_str(), tlafile.c_str(), depfile.c_str(), cinfile.c_str(), coutfile.c_str(),
                                               datasrc.c_str(), lyrname.c_str(), uselyrname, lyrno, useOutlets, usec, contcheck);
    return cli::finish("tlaccum", err);
}
