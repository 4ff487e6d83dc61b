// sdp_kernel_args.h -- plain-old-data argument blocks shared by the host
// library (sdp_hip.hip) and the device kernels (built-in and generated-model
// code objects).  Passed by value as the single kernel argument.
#pragma once
#include <stdint.h>

#define SDP_MAXD 4   // multilinear_cython.pyx:33-47 dispatches d = 1..4 only

// Every model code object carries `extern "C" __constant__ int32_t sdp_meta[SDP_META_WORDS]`: what it
// was generated for.  sdp_problem_create reads it and refuses (SDP_EMODULE) a problem it does not
// match -- compiled-in table sizes, dtype, dimensions -- instead of launching a kernel that would
// return stale results (the kernels trap on such a launch: sdp_trap_unless).
#define SDP_META_WORDS 16
#define SDP_META_MAGIC 0x33504453      // "SDP3"
enum {
    SDP_META_MAGIC_AT = 0, SDP_META_REAL_BYTES, SDP_META_D, SDP_META_NU, SDP_META_HAS_W,
    SDP_META_LAYOUT,        // 0 node order, 1 columns (axis 0 fastest)
    SDP_META_COL_N0,        // points of axis 0 the column table was sized for (0: node-order unit)
    SDP_META_COL_W,         // perturbation points of the column table (>= 1)
    SDP_META_FLAGS,         // SDP_META_F_*
    SDP_META_UTAB,          // tabulated values per control (0: none)
    SDP_META_UTAB_N,        // controls the control table has room for (the lattice may be shorter)
    SDP_META_THREADS,       // workgroup size of the sweep kernel (column / staged units)
    SDP_META_COL_ROWS,      // rows of axis 0 the table holds (< COL_N0: row window)
    SDP_META_LEAD_AXES,     // SDP_META_F_LEAD units: controlled state variables (the plane-major arrays' "lead" axes)
    SDP_META_LEAD_PERM,     // ... and which state variable logical axis j is (nibble j; stocks first)
    SDP_META_TAIL_BYTES     // resident-chunk column kernel that keeps its tail in global memory (SdpSweepArgs.tail): bytes per workgroup
};
enum {
    SDP_META_F_FILTER = 1, SDP_META_F_WINDOW = 2, SDP_META_F_TRAIL_HAS_U = 4, SDP_META_F_STAGED = 8,
    SDP_META_F_WPAIR = 16, SDP_META_F_LEAN = 32,
    SDP_META_F_CLAIMS = 64, // the sweep kernel's workgroups claim their units (persistent: a bounded grid)
    SDP_META_F_SHIFT = 128, // certified filter on the shifted lattice (a perturbation that reaches x0')
    SDP_META_F_LEAD = 256,  // node-order sweep with the filter on an array reduced over w (sdp_lead_kernel.h):
                            // the code object also exports sdp_lead_reduce, launched before every sweep
    SDP_META_F_PEER_STORES = 512   // the backup kernels store J through sdp_store_J (SdpSweepArgs.peer_J: direct exchange)
};
#define SDP_MAX_PEERS 8     // ranks of a direct exchange: the GPUs of one node
#define SDP_MAXU 4   // control variables per system

// One Bellman backup over a contiguous range of state nodes
// (reference stodynprog.py:466-534 + 639-691).
struct SdpSweepArgs {
    const void *V;         // [S] cost-to-go J_next, C-order, last axis fastest
    void *J;               // [S] output J_k (only nodes in [node_begin,node_end) written)
    void *pol;             // [S*nu] output optimal control VALUES (may be null)
    int32_t *idx;          // [S] output flat C-order index into the control lattice (may be null)
    const void *axes;      // concatenated state-grid axes (reals); axis k at axis_off[k]
    const void *wgrid;     // [W] perturbation grid (null when W == 0)
    const void *proba;     // [W] perturbation weights
    const void *box_lo;    // control box lower ends: [nu] or [nu][S] (per node)
    const void *box_hi;    // control box upper ends
    const int32_t *box_n;  // number of control points: [nu] or [nu][S]
    const void *pol_in;    // [S*nu] prescribed policy (eval_policy kernel only)
    int64_t node_begin;    // first node of this launch (flat C-order id)
    int64_t node_end;      // one past the last node
    int64_t S;             // total number of state nodes
    double t_k;            // time index for non-stationary systems
    int32_t orders[SDP_MAXD];
    int32_t axis_off[SDP_MAXD];
    int32_t W;             // perturbation points (0: deterministic system)
    int32_t box_per_node;  // 0: constant box, 1: per-node arrays
    // ---- column layout only (sdp_column_kernel.h) ----------------------------
    // Arrays over nodes (V, J, pol, idx, box_*, pol_in) are then stored with the
    // LEADING state axis fastest: element (c, i) at c*n_lead + i, c = C-order
    // index over axes 1..d-1 ("column"), i = index along axis 0.
    int64_t col_begin;     // first column of this launch
    int64_t col_end;       // one past the last column
    int32_t n_lead;        // orders[0]
    int32_t col_splits;    // workgroups sharing one column (each redoes the table)
    // ---- policy-evaluation kernels only: fused relative-DP shift ------------------
    int64_t shift_index;   // >= 0: every V read is V[.] - V[shift_index] (device order); -1: none
    double *ref_out;       // if set, thread 0 of workgroup 0 stores V[shift_index] there (J_ref of the previous step)
    // ---- diagnostic builds only (SDP_STAMP code objects; null in production) ----
    // [gridDim.x][4] words: s_memtime / s_memrealtime of thread 0 at kernel entry and exit
    // (in-kernel clock = d memtime / d memrealtime x 100 MHz, MI355X_MICROARCH.md DVFS item 6)
    unsigned long long *stamps;
    // ---- filtered column kernel: units handed out in order (sdp_column_kernel.h) ----
    // claim[32 * k], k = 0..7: next unit of XCD k's share (relative); claim[256]: workgroups that
    // have finished (the last one zeroes everything for the next launch).  Zero before the first launch.
    unsigned int *claim;
    // ---- several controlled state variables (sdp_lead_kernel.h): the array reduced over w, in node order ----
    void *aux_a;           // [S] A = sum_w p_w inner_w, plane-major, written by sdp_lead_reduce, read by sdp_sweep
    void *aux_v;           // [S] copy of V, plane-major (the second pass reads it)
    void *aux_e;           // [nodes per block of trailing coordinates] bound factor of the trailing cells
    unsigned long long *aux_vmax;   // bits of max |V| as a double (zero before sdp_lead_reduce)
    int64_t aux_begin;     // plane-major arrays: lead indices [aux_begin, aux_end) hold valid values (the part of the
    int64_t aux_end;       //   grid sdp_lead_reduce was run on); a node whose controls reach outside takes the long way on V
    // ---- direct exchange (several GPUs, sdp_problem_set_direct_exchange): the kernel that computes J[node] also
    // stores it into the J buffers of the other ranks, mapped into this process (HIP IPC; xGMI stores) ----
    void *peer_J[SDP_MAX_PEERS];        // null: no store (this rank itself, ranks beyond the communicator)
    const unsigned char *peer_mask;     // column layout: [columns] bit q set = rank q reads the column; null: every peer gets every node
    int32_t n_peer;                     // 0: single GPU or another exchange -- no peer stores at all
    int32_t pad_;
    // ---- resident-chunk column kernel (sdp_colres_kernel.h, SDP_COL_TAIL_KEEP): the tail of a workgroup's table, written by
    // its first build and read back by the second pass instead of building it again ----
    void *tail;                         // [gridDim.x][sdp_meta[SDP_META_TAIL_BYTES]] bytes; private to a workgroup, no initial contents
};

// Batched closed-loop simulation (the user loop of the reference's examples, e.g.
// examples/20 Searev storage control/storage_control.py:242-251): per step the
// policy is looked up by multilinear interpolation and the dynamics advance.
struct SdpSimArgs {
    const void *pol;       // [nu][S] policy (control VALUES on the state grid), C order
    const void *axes;      // concatenated state-grid axes, like SdpSweepArgs
    const void *x0;        // [d][B] start states
    const void *w;         // [T][B] perturbation sequences (null: deterministic system)
    void *x;               // [T+1][d][B] states (x[0] = x0)
    void *u;               // [T][nu][B] controls applied
    void *g;               // [T][B] instantaneous costs (may be null)
    int64_t B;             // trajectories
    int64_t T;             // steps
    int64_t S;             // state nodes
    double t0;             // time index of step 0 (non-stationary systems)
    int32_t orders[SDP_MAXD];
    int32_t axis_off[SDP_MAXD];
};

// Stand-alone multilinear interpolation (multilinear_cython.pyx:17-49).
struct SdpInterpArgs {
    const void *values;    // [n_v][S]
    const void *s;         // [d][n_s] query coordinates
    void *out;             // [n_v][n_s]
    int64_t n_s;
    int64_t S;
    int32_t n_v;
    int32_t d;
    int32_t orders[SDP_MAXD];
    double smin[SDP_MAXD]; // converted to the kernel's real type on device
    double smax[SDP_MAXD];
};

// Tabulated backup: the model callbacks were evaluated on the host
// (stodynprog.py:674,676); the device does gather + expectation + argmin.
struct SdpTabArgs {
    const void *V;          // [S]
    const void *x_next;     // [d][n_cells]
    const void *g;          // [n_cells]
    const int64_t *cell_off;// [n_nodes+1] first cell of each node
    const void *proba;      // [W]
    void *J;                // [n_nodes]
    int32_t *idx;           // [n_nodes]
    int64_t n_nodes;
    int64_t n_cells;
    int32_t W;
    int32_t d;
    int32_t orders[SDP_MAXD];
    double smin[SDP_MAXD];
    double smax[SDP_MAXD];
};
