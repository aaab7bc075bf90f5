// threshold -ssa ssa -src src [-mask m] [-thresh t]   (flag surface of src/Thresholdmn.cpp:50-130; default threshold 100)
#include "cli_common.hpp"

static void usage(const char* prog) {
    printf("Simple Use:\n %s <basefilename>\n", prog);
    printf("Use with specific file names:\n %s -ssa <ssafile> -src <srcfile> [-mask <maskfile>] [-thresh <threshold>]\n", prog);
    printf("  <ssafile>   the grid to be thresholded (e.g. a contributing area), input\n");
    printf("  <srcfile>   1 where ssa >= threshold (and mask >= 0), 0 elsewhere, no data where ssa has no data (output)\n");
    printf("  <threshold> default 100\n");
    printf("With the simple form the suffixes ssa and src are inserted before the extension of <basefilename>.\n");
    exit(0);
}

int main(int argc, char** argv) {
    cli::take_gpus(argc, argv);
    std::string ssafile, srcfile, maskfile, tval;
    int usemask = 0;
    float thresh = 100.f;
    if (argc < 2) usage(argv[0]);
    if (argc == 2) { ssafile = cli::nameadd(argv[1], "ssa"); srcfile = cli::nameadd(argv[1], "src"); }
    cli::Args a(argc, argv);
    while (a.more()) {
        if (a.is("-ssa")) { if (!a.value(ssafile)) usage(argv[0]); }
        else if (a.is("-src")) { if (!a.value(srcfile)) usage(argv[0]); }
        else if (a.is("-mask")) { if (!a.value(maskfile)) usage(argv[0]); usemask = 1; }
        else if (a.is("-thresh")) { if (!a.value(tval)) usage(argv[0]); sscanf(tval.c_str(), "%f", &thresh); }
        else usage(argv[0]);
    }
    const int err = tdx_tool_threshold(ssafile.c_str(), srcfile.c_str(), maskfile.c_str(), thresh, usemask);
    if (err != 0 && !cli::is_abort_code(err)) { printf("Threshold Error %d\n", err); return 0; }
    return cli::finish("Threshold", err);
}
