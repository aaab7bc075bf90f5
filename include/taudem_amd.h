/* taudem_amd - C ABI of the MI355X (gfx950) implementation of TauDEM's D8 / D-infinity
 * flow-direction and contributing-area hot path.
 *
 * TauDEM (dtarb/TauDEM 5.4.0) has no library or FFI surface: its boundary is one C++ function per
 * command-line tool that reads rasters, computes and writes rasters.  This header exports that
 * boundary twice:
 *
 *   (1) in-memory entry points (tdx_<tool> / tdx_<tool>_dev) that replace the COMPUTE part of each
 *       reference tool function - the region between "dem.read(...)" and "fel.write(...)":
 *         tdx_pitremove        <- flood()     src/flood.cpp:132-482   (decl. src/flood.h:1-2)
 *         tdx_d8flowdir        <- setdird8()  src/d8.cpp:227-320      (decl. src/d8.h:5)
 *         tdx_aread8           <- aread8()    src/aread8.cpp:175-307  (decl. src/aread8.h:3)
 *         tdx_dinfflowdir      <- setdir()    src/dinf.cpp:156-243    (decl. src/tardemlib.h:70)
 *         tdx_areadinf         <- area()      src/areadinf.cpp:138-268 (decl. src/areadinf.h:2)
 *         tdx_dinfdecayaccum   <- dmarea()    src/dinfdecayaccum.cpp:165-294 (:61-62)
 *       Rasters are row-major, x fastest, row 0 = north (src/linearpart.h:506); dtypes and nodata
 *       conventions are the reference's (SURVEY.md 8b "Output conventions").
 *       The *_dev variants take DEVICE pointers (HBM-resident inputs/outputs, the benchmark path);
 *       the plain variants take HOST pointers and do the PCIe copies themselves.
 *
 *   (2) file-level entry points (tdx_tool_*) with the argument lists of the reference's tool
 *       functions (same order and meaning, `bool` spelled `int`), reading and writing GeoTIFF, so
 *       the reference's *mn.cpp mains (and the ArcGIS/python wrappers that shell out to them) can
 *       bind to this library unchanged.  See INTEGRATION.md.
 *
 * Error convention follows the reference: 0 = ok, 1 = input rasters do not match
 * (src/flood.cpp:83, src/aread8.cpp:168); the reference's MPI_Abort codes are returned instead of
 * aborting (21 file open src/tiffIO.cpp:69, 22 no writable driver src/tiffIO.cpp:313, 5 outlets
 * src/aread8.cpp:125); library-specific failures are negative.  tdx_last_error() gives the text.
 */
#ifndef TAUDEM_AMD_H
#define TAUDEM_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDX_OK 0
#define TDX_ERR_MISMATCH 1
#define TDX_ERR_OUTLETS 5
#define TDX_ERR_FILE 21
#define TDX_ERR_DRIVER 22
#define TDX_ERR_ARG (-1)
#define TDX_ERR_HIP (-2)
#define TDX_ERR_NOGPU (-3)
#define TDX_ERR_VERIFY (-4) /* TDX_SWEEP_VERIFY=1: a swept cell does not follow from its contributors' final records */
#define TDX_ERR_NOMEM (-999) /* src/linearpart.h:155 */

/* nodata conventions of the reference's outputs */
#define TDX_FEL_NODATA (-3.0e38f)      /* src/flood.cpp:136 */
#define TDX_P_NODATA ((int16_t)-32768) /* src/d8.cpp:231 */
#define TDX_SLOPE_NODATA (-1.0f)       /* src/d8.cpp:278 */
#define TDX_AREA_NODATA (-1.0f)        /* src/aread8.cpp:193,310 */
#define TDX_ANG_NODATA (-3.402823466e+38f) /* MISSINGFLOAT = -FLT_MAX, src/commonLib.h:80 */

typedef struct tdx_context tdx_context;

/* Per-call statistics (all optional: pass NULL).  Times are milliseconds measured with HIP events
 * on the context's stream; counts mirror what the reference prints to stderr. */
typedef struct tdx_stats {
    double ms_total;        /* whole call, device side (first enqueue .. last completion)          */
    double ms_kernel[8];    /* per kernel class, see TDX_K_* below                                   */
    int64_t launches[8];    /* launches per kernel class                                             */
    int64_t rounds;         /* pitremove: relaxation rounds; aread*: outer rounds                    */
    int64_t flats_initial;  /* "All slopes evaluated. %ld flats to resolve."  src/d8.cpp:296         */
    int64_t flats_left;     /* flats remaining after the last resolveflats iteration                 */
    int64_t flat_iterations;/* calls of resolveflats  src/d8.cpp:305-316                             */
    int64_t levels_fall;    /* BFS levels of incfall (sum over iterations)                           */
    int64_t levels_rise;    /* BFS levels of incrise (sum over iterations)                           */
    int64_t cells_evaluated;/* aread8/areadinf/decay: cells that received a value                    */
    int64_t levels_fall_max;/* largest incfall level of any one iteration (all strips): the level fields are int16 like the */
    int64_t levels_rise_max;/* reference's elev2 / dn partitions (src/d8.cpp:483,486); a flat deeper than 32766 levels - where */
                            /* the reference's short counters wrap - restarts the call on int32 fields                       */
} tdx_stats;

/* kernel classes for ms_kernel[] / launches[] */
#define TDX_K_STENCIL 0  /* streaming 3x3 stencils: seeds, slope pass, in-degree                      */
#define TDX_K_RELAX 1    /* pitremove tile relaxation                                                  */
#define TDX_K_BFS 2      /* flat resolution frontier sweeps                                            */
#define TDX_K_FLATDIR 3  /* setFlow2 / SET2 on flats + bookkeeping                                     */
#define TDX_K_ACCUM 4    /* dependency-driven accumulation sweep                                       */
#define TDX_K_MISC 5     /* fills, compaction, copies                                                  */
#define TDX_K_TILEK 6    /* the tile-relaxation kernel alone (nested inside RELAX / BFS): per-launch durations */

/* ---- context ------------------------------------------------------------------------------ */
/* Creates a context bound to HIP device `device` (one context per GPU / per process rank).
 * Fails with TDX_ERR_NOGPU when no HIP device is available: there is no CPU fallback. */
int tdx_context_create(int device, tdx_context** ctx);
void tdx_context_destroy(tdx_context* ctx);
/* frees the context's scratch arena (it grows again on demand): for hosts that keep one context across workloads of very different sizes */
int tdx_context_release_scratch(tdx_context* ctx);
const char* tdx_last_error(const tdx_context* ctx); /* ctx may be NULL: last error of the calling thread */
int tdx_synchronize(tdx_context* ctx);
void* tdx_stream(tdx_context* ctx);                  /* the hipStream_t all work is enqueued on */
const char* tdx_version(void);
/* Context options.  "kernel_timing" (0/1, default 0): additionally bracket every launch of the tile-relaxation kernel
 * with HIP events (TDX_K_TILEK in tdx_stats) - two event records per launch, so it is off unless asked for. */
int tdx_context_set_option(tdx_context* ctx, const char* name, int64_t value);
int tdx_device_count(void);
/* halo exchanges and all-reduces this context has taken part in since it was created (strip runs; 0 on a single strip) */
void tdx_context_comm_counters(const tdx_context* ctx, int64_t* exchanges, int64_t* allreduces);
/* Segment trace (option "segment_trace" = 1 | 2): a strip call is a sequence of segments of rank-local work, each ended by a collective (the
 * role of share() / MPI_Allreduce in src/linearpart.h:194-360, src/aread8.cpp:282-303) or by the end of the call; every rank passes through the
 * same sequence.  device_ms = HIP-event time of the segment on the rank's stream, wall_ms = host time.  Mode 2 lets one rank at a time onto the
 * device (ranks that share a GPU), so that a segment is timed as it would run on a GPU of its own: sum over segments of the maximum over ranks
 * + collectives x latency is the critical path of the N-GPU run (scripts/project_8gpu.py).  tdx_context_segments copies up to `capacity` records,
 * clears the log when `out` is given, and returns the number of records that were logged. */
typedef struct tdx_segment {
    char stage[24];   /* "pitremove", "d8flowdir", "aread8" ...                                  */
    char phase[24];   /* sub-stage, free text ("forest", "big cells"); may be empty              */
    int32_t kind;     /* what ended the segment: 0 halo exchange, 1 all-reduce, 2 end of the call */
    float device_ms;
    float wall_ms;
} tdx_segment;
int64_t tdx_context_segments(tdx_context* ctx, tdx_segment* out, int64_t capacity);

/* device memory helpers so that callers without a HIP binding (ctypes, cgo ...) can stage data */
int tdx_device_alloc(tdx_context* ctx, uint64_t bytes, void** dptr);
int tdx_device_free(tdx_context* ctx, void* dptr);
int tdx_copy_to_device(tdx_context* ctx, void* dptr, const void* host, uint64_t bytes);
int tdx_copy_to_host(tdx_context* ctx, void* host, const void* dptr, uint64_t bytes);

/* ---- PitRemove ---------------------------------------------------------------------------- */
/* fel = pit-filled dem.  mask: optional depression mask (cells == 1 keep their elevation,
 * src/flood.cpp:249), NULL if unused.  fourway != 0 = the -4way flag (src/flood.cpp:68-70). */
int tdx_pitremove_dev(tdx_context* ctx, const float* d_dem, int64_t nx, int64_t ny, float dem_nodata,
                      const int16_t* d_mask, int fourway, float* d_fel, tdx_stats* stats);
int tdx_pitremove(tdx_context* ctx, const float* dem, int64_t nx, int64_t ny, float dem_nodata,
                  const int16_t* mask, int fourway, float* fel, tdx_stats* stats);

/* ---- D8FlowDir ---------------------------------------------------------------------------- */
/* dxc, dyc: HOST arrays of ny per-row cell sizes in metres (src/linearpart.h:516-534);
 * p: int16 D8 codes 1..8, 0 unresolved flat, -32768 nodata; sd8: slope, -1 nodata (may be NULL). */
int tdx_d8flowdir_dev(tdx_context* ctx, const float* d_fel, int64_t nx, int64_t ny, float fel_nodata,
                      const double* dxc, const double* dyc, int16_t* d_p, float* d_sd8, tdx_stats* stats);
int tdx_d8flowdir(tdx_context* ctx, const float* fel, int64_t nx, int64_t ny, float fel_nodata,
                  const double* dxc, const double* dyc, int16_t* p, float* sd8, tdx_stats* stats);

/* ---- AreaD8 ------------------------------------------------------------------------------- */
/* w: optional weight grid (NULL = unit weights).  contcheck: 1 = edge contamination (default),
 * 0 = the -nc flag.  Outlets: n_outlets < 0 = no outlets; otherwise HOST arrays of global
 * column/row indices as produced by tiffIO::geoToGlobalXY (src/tiffIO.cpp:580-588). */
int tdx_aread8_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata,
                   const float* d_w, float w_nodata, int contcheck,
                   const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                   float* d_ad8, tdx_stats* stats);
int tdx_aread8(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata,
               const float* w, float w_nodata, int contcheck,
               const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
               float* ad8, tdx_stats* stats);

/* ---- GridNet / Threshold (the step after AreaD8) ------------------------- */
/* gridnet() src/gridnet.cpp:54-514: plen / tlen = longest / total upstream path length (float, -1 nodata), gord = Strahler order
 * (int16, -1 nodata).  mask: optional int32 grid, only cells with mask >= thresh are evaluated (NULL = all cells).  dxc / dyc: per-row
 * cell sizes (ny doubles, HOST).  Outlets (src/gridnet.cpp:269-369): n_outlets < 0 = none; otherwise HOST arrays of global column / row
 * indices - only the outlets' upstream closure is evaluated, other cells get gord 0.  Strips: as tdx_aread8_strip (mask strip halo rows
 * are filled by the library; outlet rows in strip-array coordinates). */
int tdx_gridnet_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata,
                    const double* dxc, const double* dyc, const int32_t* d_mask, int32_t thresh,
                    const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                    float* d_plen, float* d_tlen, int16_t* d_gord, tdx_stats* stats);
int tdx_gridnet(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata,
                const double* dxc, const double* dyc, const int32_t* mask, int32_t thresh,
                const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                float* plen, float* tlen, int16_t* gord, tdx_stats* stats);
/* threshold() src/Threshold.cpp:49-162: src = 1 where ssa >= thresh (and mask >= 0), else 0; -32768 where ssa is nodata */
int tdx_threshold_dev(tdx_context* ctx, const float* d_ssa, int64_t nx, int64_t ny, float ssa_nodata,
                      const float* d_mask, float thresh, int16_t* d_src, tdx_stats* stats);
int tdx_threshold(tdx_context* ctx, const float* ssa, int64_t nx, int64_t ny, float ssa_nodata,
                  const float* mask, float thresh, int16_t* src, tdx_stats* stats);

/* d8flowpathextremeup() src/D8flowpathextremeup.cpp:58-285: ssa = maximum (usemax = 1) or minimum of the grid sa over everything
 * upstream of a cell along D8 flow paths (float, nodata -FLT_MAX).  contcheck / outlets as for AreaD8; strips: as tdx_aread8_strip. */
int tdx_d8flowpathextremeup_dev(tdx_context* ctx, const int16_t* d_p, int64_t nx, int64_t ny, int16_t p_nodata,
                                const float* d_sa, int usemax, int contcheck,
                                const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                                float* d_ssa, tdx_stats* stats);
int tdx_d8flowpathextremeup(tdx_context* ctx, const int16_t* p, int64_t nx, int64_t ny, int16_t p_nodata,
                            const float* sa, int usemax, int contcheck,
                            const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                            float* ssa, tdx_stats* stats);

/* ---- DinfFlowDir -------------------------------------------------------------------------- */
int tdx_dinfflowdir_dev(tdx_context* ctx, const float* d_fel, int64_t nx, int64_t ny, float fel_nodata,
                        const double* dxc, const double* dyc, float* d_ang, float* d_slp, tdx_stats* stats);
int tdx_dinfflowdir(tdx_context* ctx, const float* fel, int64_t nx, int64_t ny, float fel_nodata,
                    const double* dxc, const double* dyc, float* ang, float* slp, tdx_stats* stats);

/* ---- AreaDinf / DinfDecayAccum ------------------------------------------------------------ */
int tdx_areadinf_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                     const double* dxc, const double* dyc, const float* d_w, int contcheck,
                     const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                     float* d_sca, tdx_stats* stats);
int tdx_areadinf(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                 const double* dxc, const double* dyc, const float* w, int contcheck,
                 const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                 float* sca, tdx_stats* stats);
int tdx_dinfdecayaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                           const double* dxc, const double* dyc, const float* d_dm, float dm_nodata,
                           const float* d_w, int contcheck,
                           const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                           float* d_dsca, tdx_stats* stats);
int tdx_dinfdecayaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                       const double* dxc, const double* dyc, const float* dm, float dm_nodata,
                       const float* w, int contcheck,
                       const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                       float* dsca, tdx_stats* stats);


/* ---- D-infinity flow algebra in reverse --------------------------------------------------------- */
/* depgrd() src/DinfUpDependence.cpp:52-272: dep = fraction of the flow of each cell that reaches the cells with dg >= 1 (the
 * "disturbance" grid, int32); float, nodata -1.  dxc / dyc: per-row cell sizes (HOST). */
int tdx_dinfupdependence_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                             const double* dxc, const double* dyc, const int32_t* d_dg, float* d_dep, tdx_stats* stats);
int tdx_dinfupdependence(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                         const double* dxc, const double* dyc, const int32_t* dg, float* dep, tdx_stats* stats);
/* dsaccum() src/DinfRevAccum.cpp:51-290: racc = weight accumulated over everything DOWNSTREAM of a cell (proportioned like the
 * flow), dmax = the maximum weight downstream; both float, nodata -FLT_MAX. */
int tdx_dinfrevaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                         const double* dxc, const double* dyc, const float* d_w, float w_nodata,
                         float* d_racc, float* d_dmax, tdx_stats* stats);
int tdx_dinfrevaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                     const double* dxc, const double* dyc, const float* w, float w_nodata,
                     float* racc, float* dmax, tdx_stats* stats);
/* dsllArea() src/DinfConcLimAccum.cpp:61-326: concentration limited accumulation.  ctpt = csol where the indicator dg (int16,
 * a SHORT grid in the reference) is > 0, else sum over the contributing cells of p * ctpt * q * dm divided by the cell's own q;
 * cells with q <= 0 get no value.  Result float, nodata -FLT_MAX.  Outlets as for tdx_areadinf. */
int tdx_dinfconclimaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                             const double* dxc, const double* dyc, const float* d_dm, float dm_nodata, const int16_t* d_dg,
                             const float* d_q, float q_nodata, float csol, int contcheck,
                             const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* d_ctpt, tdx_stats* stats);
int tdx_dinfconclimaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                         const double* dxc, const double* dyc, const float* dm, float dm_nodata, const int16_t* dg,
                         const float* q, float q_nodata, float csol, int contcheck,
                         const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets, float* ctpt, tdx_stats* stats);
/* tlaccum() src/DinfTransLimAccum.cpp:61-372: transport limited accumulation.  tla = min(inflow + tsup, tc), tdep = what stays
 * behind; with a concentration grid cs (and then ctpt != NULL) the concentration of the transported material as well.  All
 * float, nodata -FLT_MAX; cs and ctpt are both NULL or both given. */
int tdx_dinftranslimaccum_dev(tdx_context* ctx, const float* d_ang, int64_t nx, int64_t ny, float ang_nodata,
                              const double* dxc, const double* dyc, const float* d_tsup, float tsup_nodata, const float* d_tc,
                              float tc_nodata, const float* d_cs, float cs_nodata, int contcheck,
                              const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                              float* d_tla, float* d_tdep, float* d_ctpt, tdx_stats* stats);
int tdx_dinftranslimaccum(tdx_context* ctx, const float* ang, int64_t nx, int64_t ny, float ang_nodata,
                          const double* dxc, const double* dyc, const float* tsup, float tsup_nodata, const float* tc,
                          float tc_nodata, const float* cs, float cs_nodata, int contcheck,
                          const int32_t* outlet_x, const int32_t* outlet_y, int64_t n_outlets,
                          float* tla, float* tdep, float* ctpt, tdx_stats* stats);

/* ---- row strips across GPUs (replaces linearpart<T>, src/linearpart.h:55-565) ---------------------- */
/* One process per GPU holds one horizontal strip of the raster plus ONE halo row above and below it,
 * exactly like linearpart (rows r*(Y/P) .. , remainder to the last rank, src/linearpart.h:133-134; halo =
 * topBorder/bottomBorder, :148-162).  A strip array therefore has (ny_local + 2) rows of nx cells:
 * row 0 = halo above, rows 1..ny_local = owned, row ny_local+1 = halo below.  The caller fills the
 * owned rows of the inputs; the library fills halos (with the neighbour's rows, or with nodata beyond the
 * global raster - the reference's borders start as nodata, :158-162).
 *
 * The library never calls a communication library itself.  It asks the host through tdx_comm:
 *   exchange(user, bytes): send send_up[0..bytes) to rank-1 and send_down[0..bytes) to rank+1; receive
 *       rank-1's send_down into recv_up and rank+1's send_up into recv_down (ends of the chain skip the
 *       missing neighbour).  Replaces share()/passBorders() (src/linearpart.h:194-299).  The four buffers
 *       are DEVICE memory owned by the host (e.g. torch tensors handed to RCCL send/recv); `capacity`
 *       bytes each, at least 16*nx.
 *   allreduce(user, values, count, op): in-place reduction of `count` int64 HOST values over all ranks,
 *       op 0 = sum, 1 = max.  Replaces MPI_Allreduce / ringTerm (src/linearpart.h:301-360).
 * Both return 0 on success and must have completed (data visible to any stream) when they return; the
 * library has synchronised its own stream before it calls them. */
typedef struct tdx_comm {
    int32_t rank, size;
    void* user;
    int (*exchange)(void* user, uint64_t bytes);
    int (*allreduce)(void* user, int64_t* values, int32_t count, int32_t op);
    void* send_up; void* send_down; void* recv_up; void* recv_down;
    uint64_t capacity;
    /* ---- optional extensions (zero / NULL = the host-synchronous contract above) ---- */
    uint64_t flags;       /* TDX_COMM_STREAM_ORDERED: exchange() ENQUEUES on the context's stream (tdx_stream) and completes in
                           * stream order; the library then neither synchronises before calling it nor expects completion on return */
    /* in-place reduction of `count` int64 DEVICE values, enqueued on the context's stream (NULL: the library copies the values
     * to the host and calls allreduce).  Lets a termination vote travel device -> RCCL -> device -> host with ONE synchronisation. */
    int (*allreduce_dev)(void* user, int64_t* d_values, int32_t count, int32_t op);
} tdx_comm;
#define TDX_OP_SUM 0
#define TDX_OP_MAX 1
#define TDX_COMM_STREAM_ORDERED 1ull

/* ---- native transports for tdx_comm (taudem_amd/csrc/comm.cpp) ------------------------------------------------------
 * (1) RCCL, one process per GPU (the launch contract of `mpiexec -n P tool`, src/linearpart.h:133-134,194-219,343-384):
 *     rank 0 calls tdx_rccl_unique_id and hands the 128 bytes to the other ranks by any means (torch.distributed store,
 *     MPI_Bcast, a file); every rank calls tdx_rccl_comm_create.  Boundary rows travel as grouped ncclSend/ncclRecv on the
 *     context's stream, votes as ncclAllReduce on device values: stream-ordered, no host round trip per exchange.
 * (2) RCCL or peer copies, ONE process driving N GPUs with one thread per GPU (what `tool --gpus N` uses): tdx_group_create
 *     builds N contexts and N communicators (ncclCommInitAll when the devices are distinct; host-synchronous peer copies
 *     between the threads' buffers when TAUDEM_AMD_COMM=peer or when several ranks share a device); tdx_group_comm(g, r) is
 *     rank r's tdx_comm, to be used from rank r's thread only. */
typedef struct tdx_rccl_comm tdx_rccl_comm;
#define TDX_RCCL_ID_BYTES 128
int tdx_rccl_unique_id(void* id128);
int tdx_rccl_comm_create(tdx_context* ctx, const void* id128, int32_t rank, int32_t size, int64_t nx, tdx_rccl_comm** out);
const tdx_comm* tdx_rccl_comm_handle(tdx_rccl_comm* c);
void tdx_rccl_comm_counters(const tdx_rccl_comm* c, int64_t* exchanges, int64_t* allreduces);
void tdx_rccl_comm_destroy(tdx_rccl_comm* c);
/* loop-back self test of the transport on one rank (send/recv to self + all-reduce): 0 = the RCCL calls work on this box */
int tdx_rccl_selftest(tdx_context* ctx);
/* latencies of the two primitives as the strip protocol uses them, microseconds per repetition (collective: every rank of `comm` calls it):
 * out_us[0] one boundary-row exchange of `bytes` per direction, out_us[1] one termination vote (all-reduce of one device int64 + read-back + wait) */
int tdx_comm_latency(tdx_context* ctx, const tdx_comm* comm, int32_t reps, uint64_t bytes, double* out_us);
/* the same for RCCL with ONE rank sending to itself (what a one-GPU box can run): a lower bound of the latencies between GPUs */
int tdx_rccl_latency(tdx_context* ctx, int32_t reps, uint64_t bytes, double* out_us);

typedef struct tdx_group tdx_group;
/* devices[size]: HIP device of every rank (repeats allowed: several ranks then share a GPU and the peer transport is used) */
int tdx_group_create(int32_t size, const int32_t* devices, int64_t nx, tdx_group** out);
tdx_context* tdx_group_context(tdx_group* g, int32_t rank);
const tdx_comm* tdx_group_comm(tdx_group* g, int32_t rank);
const char* tdx_group_transport(const tdx_group* g);   /* "rccl" | "peer" */
/* a rank failed outside a collective: the other ranks' collectives fail at once instead of waiting for it (then destroy the group) */
void tdx_group_abort(tdx_group* g);
void tdx_group_destroy(tdx_group* g);

/* Strip variants of the device entry points (comm == NULL or comm->size == 1: a single strip whose halo
 * rows lie outside the raster).  All raster pointers are DEVICE strip arrays of (ny_local + 2) x nx. */
int tdx_pitremove_strip(tdx_context* ctx, const tdx_comm* comm, float* d_dem, int64_t nx, int64_t ny_local, float dem_nodata,
                        const int16_t* d_mask, int fourway, float* d_fel, tdx_stats* stats);
/* dxc/dyc: per-row cell sizes of the ny_local + 2 strip rows */
int tdx_d8flowdir_strip(tdx_context* ctx, const tdx_comm* comm, float* d_fel, int64_t nx, int64_t ny_local, float fel_nodata,
                        const double* dxc, const double* dyc, int16_t* d_p, float* d_sd8, tdx_stats* stats);
/* D-infinity on strips: angles / slopes, then contributing area and decayed accumulation; d_w optional weight strip;
 * outlets as for tdx_aread8_strip; dxc/dyc: per-row cell sizes of the ny_local + 2 strip rows */
int tdx_dinfflowdir_strip(tdx_context* ctx, const tdx_comm* comm, float* d_fel, int64_t nx, int64_t ny_local, float fel_nodata,
                          const double* dxc, const double* dyc, float* d_ang, float* d_slp, tdx_stats* stats);
int tdx_areadinf_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                       const double* dxc, const double* dyc, const float* d_w, int contcheck, const int32_t* outlet_x, const int32_t* outlet_row,
                       int64_t n_outlets, float* d_sca, tdx_stats* stats);
int tdx_dinfdecayaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                             const double* dxc, const double* dyc, float* d_dm, float dm_nodata, const float* d_w, int contcheck,
                             const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets, float* d_dsca, tdx_stats* stats);
/* AreaD8 on a strip, optionally with weights and/or outlets (unit weights take the tile-contraction sweep).  outlet_x / outlet_row: HOST arrays in STRIP-ARRAY coordinates (row 1 = first owned
 * row; outlets outside the owned rows are ignored, like isInPartition, src/commonLib.cpp:289-291); n_outlets < 0 = none */
int tdx_aread8_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local, int16_t p_nodata,
                        const float* d_w, float w_nodata, int contcheck, const int32_t* outlet_x, const int32_t* outlet_row,
                        int64_t n_outlets, float* d_ad8, tdx_stats* stats);
int tdx_d8flowpathextremeup_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local,
                                  int16_t p_nodata, const float* d_sa, int usemax, int contcheck,
                                  const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets,
                                  float* d_ssa, tdx_stats* stats);
/* the reverse D-infinity tools on strips (dxc / dyc: per-row cell sizes of the ny_local + 2 strip rows) */
int tdx_dinfupdependence_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                               const double* dxc, const double* dyc, int32_t* d_dg, float* d_dep, tdx_stats* stats);
int tdx_dinfrevaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                           const double* dxc, const double* dyc, float* d_w, float w_nodata, float* d_racc, float* d_dmax, tdx_stats* stats);
/* the limited D-infinity accumulations on strips (outlet_row: array row of the strip, as for tdx_areadinf_strip) */
int tdx_dinfconclimaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                               const double* dxc, const double* dyc, const float* d_dm, float dm_nodata, const int16_t* d_dg,
                               const float* d_q, float q_nodata, float csol, int contcheck,
                               const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets, float* d_ctpt, tdx_stats* stats);
int tdx_dinftranslimaccum_strip(tdx_context* ctx, const tdx_comm* comm, float* d_ang, int64_t nx, int64_t ny_local, float ang_nodata,
                                const double* dxc, const double* dyc, const float* d_tsup, float tsup_nodata, const float* d_tc,
                                float tc_nodata, const float* d_cs, float cs_nodata, int contcheck,
                                const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets,
                                float* d_tla, float* d_tdep, float* d_ctpt, tdx_stats* stats);
/* GridNet on a strip (d_mask: optional int32 strip array whose halo rows the library fills) */
int tdx_gridnet_strip(tdx_context* ctx, const tdx_comm* comm, int16_t* d_p, int64_t nx, int64_t ny_local, int16_t p_nodata,
                      const double* dxc, const double* dyc, int32_t* d_mask, int32_t thresh,
                      const int32_t* outlet_x, const int32_t* outlet_row, int64_t n_outlets,
                      float* d_plen, float* d_tlen, int16_t* d_gord, tdx_stats* stats);

/* ---- synthetic benchmark input (not in the reference) -------------------------------------- */
/* Fills d_out (nx*ny float32) with the seeded fractal surface of taudem_amd/csrc/synth_dem.h for
 * the window whose top-left global cell is (x0,y0).  Bit-identical to the host generator. */
int tdx_synth_dem_dev(tdx_context* ctx, uint64_t seed, int64_t nx, int64_t ny, int64_t x0, int64_t y0,
                      int64_t base_wavelength, float* d_out);

/* ---- raster files (GeoTIFF / BigTIFF), host side -------------------------------------------- */
typedef struct tdx_raster_info {
    int64_t nx, ny;
    double geotransform[6];
    double nodata;        /* -9999 when the file declares none (src/tiffIO.cpp:161-167) */
    int32_t has_nodata;
    int32_t geographic;   /* 1: per-row dxc/dyc follow src/tiffIO.cpp:127-143 */
    double dxA, dyA;      /* src/tiffIO.cpp:155-156 */
} tdx_raster_info;
#define TDX_DT_I16 0
#define TDX_DT_I32 1
#define TDX_DT_F32 2
int tdx_raster_info_read(const char* path, tdx_raster_info* info);
/* data: nx*ny of `dtype` (converted like GDALRasterIO); dxc/dyc: optional ny doubles each */
int tdx_raster_read(const char* path, int dtype, void* data, double* dxc, double* dyc);
/* georef_from: optional path of the raster whose geotransform/projection are copied
 * (src/tiffIO.cpp:344-349).  lzw != 0 writes COMPRESS=LZW like the reference. */
int tdx_raster_write(const char* path, int dtype, const void* data, int64_t nx, int64_t ny, double nodata,
                     const char* georef_from, int lzw);
/* variant that creates georeferencing from scratch (synthetic inputs) */
int tdx_raster_write_geo(const char* path, int dtype, const void* data, int64_t nx, int64_t ny, double nodata,
                         const double* geotransform, int geographic, int lzw);

/* ---- outlet points (replaces readoutlets(), src/ReadOutlets.cpp:49-198, without OGR) -------------------------------- */
/* Reads point features from an ESRI shapefile (.shp: Point / PointZ / PointM), a GeoJSON file (.json / .geojson: Point
 * geometries) or a text file ("x y [id]" per line).  x / y / id: caller arrays of `capacity` entries (may be NULL to count);
 * *count = number of points in the file.  Returns 0, or TDX_ERR_OUTLETS (5, the code of src/aread8.cpp:125). */
int tdx_outlets_read(const char* path, double* x, double* y, int32_t* id, int64_t capacity, int64_t* count);
/* tiffIO::geoToGlobalXY (src/tiffIO.cpp:580-588) for the raster `rasterpath`: (int) truncation, no bounds check */
int tdx_outlets_to_cells(const char* rasterpath, const double* x, const double* y, int64_t n, int32_t* col, int32_t* row);

/* ---- file-level tool functions (argument lists of the reference's tool functions) ---------- */
/* int flood(char*,char*,char*,int,bool,bool,bool,char*)            src/flood.h:1-2 */
int tdx_tool_pitremove(const char* demfile, const char* felfile, const char* sfdrfile, int usesfdr,
                       int verbose, int is_4Point, int use_mask, const char* maskfile);
/* int setdird8(char*,char*,char*,char*,int)                        src/d8.h:5 */
int tdx_tool_d8flowdir(const char* demfile, const char* pointfile, const char* slopefile,
                       const char* flowfile, int useflowfile);
/* int aread8(char*,char*,char*,char*,int,int,char*,int,int,int)    src/aread8.h:3 */
int tdx_tool_aread8(const char* pfile, const char* afile, const char* datasrc, const char* lyrname,
                    int uselyrname, int lyrno, const char* wfile, int useOutlets, int usew, int contcheck);
/* int setdir(char*,char*,char*,char*,int)                          src/dinf.cpp:109 */
int tdx_tool_dinfflowdir(const char* demfile, const char* angfile, const char* slopefile,
                         const char* flowfile, int useflowfile);
/* int area(char*,char*,char*,char*,int,int,char*,int,int,int)      src/areadinf.h:2 */
int tdx_tool_areadinf(const char* angfile, const char* scafile, const char* datasrc, const char* lyrname,
                      int uselyrname, int lyrno, const char* wfile, int useOutlets, int usew, int contcheck);
/* int dmarea(char*,char*,char*,char*,char*,int,int,char*,int,int,int) src/dinfdecayaccum.cpp:61-62 */
int tdx_tool_dinfdecayaccum(const char* angfile, const char* adecfile, const char* dmfile, const char* datasrc,
                            const char* lyrname, int uselyrname, int lyrno, const char* wfile,
                            int useOutlets, int usew, int contcheck);
/* int gridnet(char*,char*,char*,char*,char*,char*,char*,int,int,int,int,int)  src/gridnet.cpp:54-55 */
int tdx_tool_gridnet(const char* pfile, const char* plenfile, const char* tlenfile, const char* gordfile,
                     const char* maskfile, const char* datasrc, const char* lyrname, int uselyrname, int lyrno,
                     int useMask, int useOutlets, int thresh);
/* int d8flowpathextremeup(char*,char*,char*,int,char*,char*,int,int,int,int)   src/D8flowpathextremeup.cpp:58 */
int tdx_tool_d8flowpathextremeup(const char* pfile, const char* safile, const char* ssafile, int usemax, const char* datasrc,
                                 const char* lyrname, int uselyrname, int lyrno, int useOutlets, int contcheck);
/* int depgrd(char* angfile, char* dgfile, char* depfile)           src/DinfUpDependence.cpp:52 */
int tdx_tool_dinfupdependence(const char* angfile, const char* dgfile, const char* depfile);
/* int dsaccum(char* angfile, char* wgfile, char* raccfile, char* dmaxfile)   src/DinfRevAccum.cpp:51 */
int tdx_tool_dinfrevaccum(const char* angfile, const char* wgfile, const char* raccfile, const char* dmaxfile);
/* int dsllArea(char* angfile, char* ctptfile, char* dmfile, char* datasrc, char* lyrname, int uselyrname, int lyrno, char* qfile,
 *              char* dgfile, int useOutlets, int contcheck, float cSol)          src/DinfConcLimAccum.cpp:61-62 */
int tdx_tool_dinfconclimaccum(const char* angfile, const char* ctptfile, const char* dmfile, const char* datasrc, const char* lyrname,
                              int uselyrname, int lyrno, const char* qfile, const char* dgfile, int useOutlets, int contcheck, float cSol);
/* int tlaccum(char* angfile, char* tsupfile, char* tcfile, char* tlafile, char* depfile, char* cinfile, char* coutfile,
 *             char* datasrc, char* lyrname, int uselyrname, int lyrno, int useOutlets, int usec, int contcheck)   src/DinfTransLimAccum.cpp:61-63 */
int tdx_tool_dinftranslimaccum(const char* angfile, const char* tsupfile, const char* tcfile, const char* tlafile, const char* depfile,
                               const char* cinfile, const char* coutfile, const char* datasrc, const char* lyrname, int uselyrname,
                               int lyrno, int useOutlets, int usec, int contcheck);
/* int threshold(char*,char*,char*,float,int)                       src/Threshold.cpp:49 */
int tdx_tool_threshold(const char* ssafile, const char* srcfile, const char* maskfile, float thresh, int usemask);
/* selects the HIP device used by the tdx_tool_* functions (default 0 / env TAUDEM_AMD_DEVICE) */
int tdx_tool_set_device(int device);
/* number of GPUs the tdx_tool_* functions partition the raster over (row strips, one thread per GPU; default 1 / env
 * TAUDEM_AMD_GPUS; the command-line tools take --gpus N).  Replaces `mpiexec -n P` (src/linearpart.h:133-134). */
int tdx_tool_set_gpus(int ngpus);

#ifdef __cplusplus
}
#endif
#endif /* TAUDEM_AMD_H */
