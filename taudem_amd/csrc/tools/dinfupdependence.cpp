// dinfupdependence -ang ang -dg dg -dep dep   (flag surface of src/DinfUpDependencemn.cpp:49-132)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -dg <dgfile> -dep <depfile>\n", prog);
    printf("  <angfile>  D-infinity flow direction input\n");
    printf("  <dgfile>   disturbance grid input (cells >= 1 are the destination zone)\n");
    printf("  <depfile>  upslope dependence output\n");
    printf("With the simple form the suffixes ang, dg and dep are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, dgfile, depfile;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-dg")) { if (!a.value(dgfile)) usage(argv[0]); }
        else if (a.is("-dep")) { if (!a.value(depfile)) usage(argv[0]); }
        else usage(argv[0]);
    }
    if (argc == 2) { angfile = cli::nameadd(argv[1], "ang"); dgfile = cli::nameadd(argv[1], "dg"); depfile = cli::nameadd(argv[1], "dep"); }
    const int err = tdx_tool_dinfupdependence(angfile.c_str(), dgfile.c_str(), depfile.c_str());
    return cli::finish("depgrd", err);
}
