// d8flowdir -fel f -p p -sd8 s [-sfdr f]   (flag surface of src/D8FlowDirmn.cpp:49-145)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <basefilename>\n", prog);
    printf("General use:\n %s -fel <demfile> -p <pointfile> -sd8 <slopefile> [-sfdr <flowfile>]\n", prog);
    printf("  <demfile>    pit-filled elevation input\n");
    printf("  <pointfile>  D8 flow direction output (1=E 2=NE 3=N 4=NW 5=W 6=SW 7=S 8=SE)\n");
    printf("  <slopefile>  D8 slope output\n");
    printf("  <flowfile>   optional existing stream raster (not supported)\n");
    printf("With the simple form the suffixes fel, p and sd8 are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string demfile, pointfile, slopefile, flowfile;
    int useflowfile = 0;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-fel")) { if (!a.value(demfile)) usage(argv[0]); }
        else if (a.is("-p")) { if (!a.value(pointfile)) usage(argv[0]); }
        else if (a.is("-sd8")) { if (!a.value(slopefile)) usage(argv[0]); }
        else if (a.is("-sfdr")) { if (!a.value(flowfile)) usage(argv[0]); useflowfile = 1; }
        else usage(argv[0]);
    }
    if (argc == 2) {
        demfile = cli::nameadd(argv[1], "fel");
        pointfile = cli::nameadd(argv[1], "p");
        slopefile = cli::nameadd(argv[1], "sd8");
    }
    const int err = tdx_tool_d8flowdir(demfile.c_str(), pointfile.c_str(), slopefile.c_str(), flowfile.c_str(), useflowfile);
    return cli::finish("d8flowdir", err);
}
