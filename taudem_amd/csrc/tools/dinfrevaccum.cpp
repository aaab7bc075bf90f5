// dinfrevaccum -ang ang -wg wg -racc racc -dmax dmax   (flag surface of src/DinfRevAccummn.cpp:49-142)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -ang <angfile> -wg <wgfile> -racc <raccfile> -dmax <dmaxfile>\n", prog);
    printf("  <angfile>   D-infinity flow direction input\n");
    printf("  <wgfile>    weight grid input\n");
    printf("  <raccfile>  reverse accumulation output\n");
    printf("  <dmaxfile>  maximum downslope output\n");
    printf("With the simple form the suffixes ang, wg, racc and dmax are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string angfile, wgfile, raccfile, dmaxfile;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ang")) { if (!a.value(angfile)) usage(argv[0]); }
        else if (a.is("-wg")) { if (!a.value(wgfile)) usage(argv[0]); }
        else if (a.is("-racc")) { if (!a.value(raccfile)) usage(argv[0]); }
        else if (a.is("-dmax")) { if (!a.value(dmaxfile)) usage(argv[0]); }
        else usage(argv[0]);
    }
    if (argc == 2) {
        angfile = cli::nameadd(argv[1], "ang"); wgfile = cli::nameadd(argv[1], "wg");
        raccfile = cli::nameadd(argv[1], "racc"); dmaxfile = cli::nameadd(argv[1], "dmax");
    }
    const int err = tdx_tool_dinfrevaccum(angfile.c_str(), wgfile.c_str(), raccfile.c_str(), dmaxfile.c_str());
    return cli::finish("dsaccum", err);
}
