// Shared pieces of the command-line mains.  The mains keep the reference's flag surface
// (src/*mn.cpp): "-flag value" pairs, one positional argument = "simple usage" where every file name
// is derived with nameadd() (src/commonLib.cpp:53-73), a usage message + exit(0) on any parse error,
// and "return 0" from main even when the tool function reports an error - except for the codes on
// which the reference calls MPI_Abort (5, 21, 22, 41-43, -999), which become the exit status.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "taudem_amd.h"

namespace cli {

// nameadd(full, arg, suff): insert suff before the extension of arg
inline std::string nameadd(const std::string& arg, const std::string& suff) {
    const size_t dot = arg.rfind('.');
    if (dot == std::string::npos) return arg + suff;
    const bool suff_has_ext = suff.rfind('.') != std::string::npos;
    return arg.substr(0, dot) + suff + (suff_has_ext ? "" : arg.substr(dot));
}

inline bool is_abort_code(int err) { return err == 5 || err == 21 || err == 22 || (err >= 41 && err <= 43) || err == -999 || err == TDX_ERR_NOGPU || err == TDX_ERR_HIP; }

// what the reference's main does after the tool function returns
inline int finish(const char* label, int err) {
    if (err == 0) return 0;
    if (is_abort_code(err)) {
        fflush(stdout);
        exit(err & 0xff);
    }
    printf("%s error %d\n", label, err);
    return 0;
}

// "--gpus N" (anywhere on the command line; not a flag of the reference, whose place for it is `mpiexec -n N`): removed from
// argv before the reference-style parsing, so that the "simple usage" test argc == 2 still works.
inline void take_gpus(int& argc, char** argv) {
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) {
            tdx_tool_set_gpus(atoi(argv[i + 1]));
            for (int j = i; j + 2 <= argc; j++) argv[j] = (j + 2 < argc) ? argv[j + 2] : nullptr;
            argc -= 2;
            i--;
        }
    }
}

struct Args {
    int argc; char** argv; int i;
    Args(int c, char** v) : argc(c), argv(v), i(c > 2 ? 1 : 2) {}
    bool more() const { return argc > i; }
    bool is(const char* flag) const { return strcmp(argv[i], flag) == 0; }
    // consumes "-flag value"; returns false when the value is missing
    bool value(std::string& out) { i++; if (argc > i) { out = argv[i]; i++; return true; } return false; }
    bool value(int& out) { i++; if (argc > i) { sscanf(argv[i], "%d", &out); i++; return true; } return false; }
    void flag() { i++; }
};

}  // namespace cli
