// taudem_amd_shim.cpp - the reference-side binding of INTEGRATION.md section 1, as a buildable file.
//
// Link the UNMODIFIED reference mains (src/PitRemovemn.cpp, src/D8FlowDirmn.cpp, src/aread8mn.cpp, src/DinfFlowDirmn.cpp,
// src/areadinfmn.cpp, src/DinfDecayAccummn.cpp, src/gridnetmn.cpp, src/Thresholdmn.cpp, src/D8FlowPathExtremeUpmn.cpp) against
// this file and -ltaudem_amd instead of flood.cpp / d8.cpp / aread8.cpp / ... + commonLib.cpp + tiffIO.cpp: every tool function the
// mains call (prototypes: src/flood.h:1-2, src/d8.h:5, src/aread8.h:3, src/tardemlib.h:70, src/areadinf.h:2,
// src/dinfdecayaccum.cpp:61-62, src/gridnet.cpp:54-55, src/Threshold.cpp:49, src/D8flowpathextremeup.cpp:58,
// src/DinfUpDependence.cpp:52, src/DinfRevAccum.cpp:51) forwards to the
// file-level C ABI, and nameadd() (src/commonLib.cpp:53-73, the only other symbol the mains use) is provided here.  No MPI and
// no GDAL at link time (their headers are only needed to COMPILE the mains, which include commonLib.h).
// oracle/Makefile builds oracle/_ref/shim_<tool> this way; tests/test_gpu_cli.py runs them against the reference's rasters.
#include <cstring>

#include "taudem_amd.h"

int flood(char* demfile, char* felfile, char* sfdrfile, int usesfdr, bool verbose, bool is_4Point, bool use_mask, char* maskfile)
{ return tdx_tool_pitremove(demfile, felfile, sfdrfile, usesfdr, verbose, is_4Point, use_mask, maskfile); }
int setdird8(char* demfile, char* pointfile, char* slopefile, char* flowfile, int useflowfile)
{ return tdx_tool_d8flowdir(demfile, pointfile, slopefile, flowfile, useflowfile); }
int aread8(char* pfile, char* afile, char* datasrc, char* lyrname, int uselyrname, int lyrno, char* wfile, int useOutlets, int usew, int contcheck)
{ return tdx_tool_aread8(pfile, afile, datasrc, lyrname, uselyrname, lyrno, wfile, useOutlets, usew, contcheck); }
int setdir(char* demfile, char* angfile, char* slopefile, char* flowfile, int useflowfile)
{ return tdx_tool_dinfflowdir(demfile, angfile, slopefile, flowfile, useflowfile); }
int area(char* angfile, char* scafile, char* datasrc, char* lyrname, int uselyrname, int lyrno, char* wfile, int useOutlets, int usew, int contcheck)
{ return tdx_tool_areadinf(angfile, scafile, datasrc, lyrname, uselyrname, lyrno, wfile, useOutlets, usew, contcheck); }
int dmarea(char* angfile, char* adecfile, char* dmfile, char* datasrc, char* lyrname, int uselyrname, int lyrno, char* wfile, int useOutlets, int usew, int contcheck)
{ return tdx_tool_dinfdecayaccum(angfile, adecfile, dmfile, datasrc, lyrname, uselyrname, lyrno, wfile, useOutlets, usew, contcheck); }
int gridnet(char* pfile, char* plenfile, char* tlenfile, char* gordfile, char* maskfile, char* datasrc, char* lyrname, int uselyrname, int lyrno, int useMask, int useOutlets, int thresh)
{ return tdx_tool_gridnet(pfile, plenfile, tlenfile, gordfile, maskfile, datasrc, lyrname, uselyrname, lyrno, useMask, useOutlets, thresh); }
int d8flowpathextremeup(char* pfile, char* safile, char* ssafile, int usemax, char* datasrc, char* lyrname, int uselyrname, int lyrno, int useOutlets, int contcheck)
{ return tdx_tool_d8flowpathextremeup(pfile, safile, ssafile, usemax, datasrc, lyrname, uselyrname, lyrno, useOutlets, contcheck); }
int depgrd(char* angfile, char* dgfile, char* depfile)
{ return tdx_tool_dinfupdependence(angfile, dgfile, depfile); }
int dsaccum(char* angfile, char* wgfile, char* raccfile, char* dmaxfile)
{ return tdx_tool_dinfrevaccum(angfile, wgfile, raccfile, dmaxfile); }
int dsllArea(char* angfile, char* ctptfile, char* dmfile, char* datasrc, char* lyrname, int uselyrname, int lyrno, char* qfile, char* dgfile, int useOutlets, int contcheck, float cSol)
{ return tdx_tool_dinfconclimaccum(angfile, ctptfile, dmfile, datasrc, lyrname, uselyrname, lyrno, qfile, dgfile, useOutlets, contcheck, cSol); }
int tlaccum(char* angfile, char* tsupfile, char* tcfile, char* tlafile, char* depfile, char* cinfile, char* coutfile, char* datasrc, char* lyrname, int uselyrname, int lyrno, int useOutlets, int usec, int contcheck)
{ return tdx_tool_dinftranslimaccum(angfile, tsupfile, tcfile, tlafile, depfile, cinfile, coutfile, datasrc, lyrname, uselyrname, lyrno, useOutlets, usec, contcheck); }
int threshold(char* ssafile, char* srcfile, char* maskfile, float thresh, int usemask)
{ return tdx_tool_threshold(ssafile, srcfile, maskfile, thresh, usemask); }

// nameadd(full, arg, suff): `suff` goes in front of the extension of `arg` (the last '.'; none: appended); a suffix that
// brings its own extension replaces the original one.  Same contract as src/commonLib.cpp:53-73.
int nameadd(char* full, char* arg, const char* suff) {
    const char* dot = strrchr(arg, '.');
    const size_t stem = dot ? size_t(dot - arg) : strlen(arg);
    memcpy(full, arg, stem);
    full[stem] = '\0';
    strcat(full, suff);
    if (dot && !strrchr(suff, '.')) strcat(full, dot);
    return int(stem);   // the reference returns the length of the stem
}
