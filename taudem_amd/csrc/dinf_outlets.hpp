// Outlets mode of the D-infinity accumulation tools (src/commonLib.cpp:165-233): the upstream closure of the outlet cells is one
// fixed point of the tile relaxation engine (flats.hpp: reach_closure); the sweep then runs on RE-CODED angles in which cells
// outside the closure keep "a valid angle" for the contamination test but neither participate nor contribute (ANG_OUTSIDE, for
// which prop() is 0 in every direction) and an outlet on a cell without angle participates as a pure sink (ANG_SINK).
// Defined in areadinf.hip; used by AreaDinf / DinfDecayAccum and by the limited accumulations of dinflim.hip.
#pragma once
#include "context.hpp"
#include "dinf_prop.hpp"
#include "strips.hpp"


// d_rows: per array row {atan2(dy, dx), dx}.  On return *ang_use points at the re-coded angles (scratch slot TDX_S_P, all rows).
int dinf_outlet_recode(tdx_context* ctx, const Strip& st, const float* d_ang, float ang_nodata, const RowProp* d_rows, const int32_t* outlet_x,
                       const int32_t* outlet_y, int64_t n_outlets, float** ang_use, tdx_stats* stats);
