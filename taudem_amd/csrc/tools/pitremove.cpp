// pitremove [-z dem] [-fel out] [-depmask mask] [-4way] [-v]   (flag surface of src/PitRemovemn.cpp:48-171)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple use:\n %s <demfile>\n", prog);
    printf("  the pit-filled output is written next to the input with 'fel' inserted before the extension;\n");
    printf("  depression masks and 4-way filling need the general form.\n\n");
    printf("General use:\n %s -z <demfile> -fel <newfile> [-depmask <maskfile>] [-4way] [-v]\n", prog);
    printf("  <demfile>   input elevation grid\n");
    printf("  <newfile>   output elevation grid with pits filled\n");
    printf("  <maskfile>  depression mask grid: cells equal to 1 keep their elevation\n");
    printf("  -4way       fill using the four edge neighbours only\n");
    printf("  -v          verbose progress messages\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string demfile, newfile, maskfile;
    bool verbose = false, is_4p = false, use_mask = false;
    if (argc < 2) { printf("Error: use either the simple form or the form with explicit file names\n"); usage(argv[0]); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-z")) { if (!a.value(demfile)) usage(argv[0]); }
        else if (a.is("-fel")) { if (!a.value(newfile)) usage(argv[0]); }
        else if (a.is("-v")) { a.flag(); verbose = true; }
        else if (a.is("-4way")) { a.flag(); is_4p = true; }
        else if (a.is("-depmask")) { if (!a.value(maskfile)) usage(argv[0]); use_mask = true; }
        else usage(argv[0]);
    }
    if (argc == 2) { demfile = argv[1]; newfile = cli::nameadd(argv[1], "fel"); }
    if (verbose) {
        printf("On input demfile: %s\nOn input newfile: %s\n", demfile.c_str(), newfile.c_str());
        printf("%ssing mask file: %s\n", use_mask ? "U" : "Not u", use_mask ? maskfile.c_str() : "N/A");
        fflush(stdout);
    }
    const int err = tdx_tool_pitremove(demfile.c_str(), newfile.c_str(), "", 0, verbose, is_4p, use_mask, maskfile.c_str());
    return cli::finish("PitRemove", err);
}
