"""File-level tool functions with the names and argument lists of the reference's tool functions
(the interface the reference's ``*mn.cpp`` mains and the ArcGIS ``pyfiles/*.py`` wrappers drive):

    flood      src/flood.h:1-2            setdird8  src/d8.h:5          aread8  src/aread8.h:3
    setdir     src/dinf.cpp:109           area      src/areadinf.h:2    dmarea  src/dinfdecayaccum.cpp:61-62

They read GeoTIFF inputs, run the HIP hot path on one GPU and write GeoTIFF outputs; return codes are
the reference's (0 ok, 1 mismatch; 21/22/5 where the reference would MPI_Abort with that code).
"""
from __future__ import annotations

from . import _lib


def _b(s):
    return (s or "").encode()


def set_device(device: int):
    return _lib.load().tdx_tool_set_device(int(device))


def flood(demfile, felfile, sfdrfile="", usesfdr=0, verbose=False, is_4Point=False, use_mask=False, maskfile=""):
    return _lib.load().tdx_tool_pitremove(_b(demfile), _b(felfile), _b(sfdrfile), int(usesfdr), int(verbose), int(is_4Point), int(use_mask), _b(maskfile))


def setdird8(demfile, pointfile, slopefile, flowfile="", useflowfile=0):
    return _lib.load().tdx_tool_d8flowdir(_b(demfile), _b(pointfile), _b(slopefile), _b(flowfile), int(useflowfile))


def aread8(pfile, afile, datasrc="", lyrname="", uselyrname=0, lyrno=0, wfile="", useOutlets=0, usew=0, contcheck=1):
    return _lib.load().tdx_tool_aread8(_b(pfile), _b(afile), _b(datasrc), _b(lyrname), int(uselyrname), int(lyrno), _b(wfile), int(useOutlets),
                                      int(usew), int(contcheck))


def setdir(demfile, angfile, slopefile, flowfile="", useflowfile=0):
    return _lib.load().tdx_tool_dinfflowdir(_b(demfile), _b(angfile), _b(slopefile), _b(flowfile), int(useflowfile))


def area(angfile, scafile, datasrc="", lyrname="", uselyrname=0, lyrno=0, wfile="", useOutlets=0, usew=0, contcheck=1):
    return _lib.load().tdx_tool_areadinf(_b(angfile), _b(scafile), _b(datasrc), _b(lyrname), int(uselyrname), int(lyrno), _b(wfile),
                                        int(useOutlets), int(usew), int(contcheck))


def dmarea(angfile, adecfile, dmfile, datasrc="", lyrname="", uselyrname=0, lyrno=0, wfile="", useOutlets=0, usew=0, contcheck=1):
    return _lib.load().tdx_tool_dinfdecayaccum(_b(angfile), _b(adecfile), _b(dmfile), _b(datasrc), _b(lyrname), int(uselyrname), int(lyrno),
                                              _b(wfile), int(useOutlets), int(usew), int(contcheck))


def gridnet(pfile, plenfile, tlenfile, gordfile, maskfile="", datasrc="", lyrname="", uselyrname=0, lyrno=0, useMask=0, useOutlets=0, thresh=0):
    return _lib.load().tdx_tool_gridnet(_b(pfile), _b(plenfile), _b(tlenfile), _b(gordfile), _b(maskfile), _b(datasrc), _b(lyrname), int(uselyrname),
                                       int(lyrno), int(useMask), int(useOutlets), int(thresh))


def d8flowpathextremeup(pfile, safile, ssafile, usemax=1, datasrc="", lyrname="", uselyrname=0, lyrno=0, useOutlets=0, contcheck=1):
    return _lib.load().tdx_tool_d8flowpathextremeup(_b(pfile), _b(safile), _b(ssafile), int(usemax), _b(datasrc), _b(lyrname), int(uselyrname), int(lyrno),
                                                   int(useOutlets), int(contcheck))


def threshold(ssafile, srcfile, maskfile="", thresh=100.0, usemask=0):
    return _lib.load().tdx_tool_threshold(_b(ssafile), _b(srcfile), _b(maskfile), float(thresh), int(usemask))


def nameadd(arg: str, suff: str) -> str:
    """nameadd() of src/commonLib.cpp:53-73: insert `suff` before the extension of `arg`."""
    dot = arg.rfind(".")
    if dot < 0:
        return arg + suff
    ext = arg[dot:]
    return arg[:dot] + suff + ("" if "." in suff else ext)


def depgrd(angfile, dgfile, depfile):
    """src/DinfUpDependence.cpp:52"""
    return _lib.load().tdx_tool_dinfupdependence(_b(angfile), _b(dgfile), _b(depfile))


def dsaccum(angfile, wgfile, raccfile, dmaxfile):
    """src/DinfRevAccum.cpp:51"""
    return _lib.load().tdx_tool_dinfrevaccum(_b(angfile), _b(wgfile), _b(raccfile), _b(dmaxfile))


def dsllArea(angfile, ctptfile, dmfile, datasrc="", lyrname="", uselyrname=0, lyrno=0, qfile="", dgfile="", useOutlets=0, contcheck=1, cSol=1.0):
    """src/DinfConcLimAccum.cpp:61"""
    return _lib.load().tdx_tool_dinfconclimaccum(_b(angfile), _b(ctptfile), _b(dmfile), _b(datasrc), _b(lyrname), int(uselyrname), int(lyrno), _b(qfile), _b(dgfile),
                                                 int(useOutlets), int(contcheck), float(cSol))


def tlaccum(angfile, tsupfile, tcfile, tlafile, depfile, cinfile="", coutfile="", datasrc="", lyrname="", uselyrname=0, lyrno=0, useOutlets=0, usec=0, contcheck=1):
    """src/DinfTransLimAccum.cpp:61"""
    return _lib.load().tdx_tool_dinftranslimaccum(_b(angfile), _b(tsupfile), _b(tcfile), _b(tlafile), _b(depfile), _b(cinfile), _b(coutfile), _b(datasrc), _b(lyrname),
                                                  int(uselyrname), int(lyrno), int(useOutlets), int(usec), int(contcheck))
