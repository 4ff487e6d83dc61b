/*
 * sdp_hip.h -- C ABI of libsdp_hip.so, the MI355X (gfx950) implementation of
 * the stodynprog value-iteration hot path.
 *
 * The reference (pierre-haessig/stodynprog) is a Python library whose only
 * native boundary is one Cython function; this header declares the C entry
 * points a binding (ctypes, cffi, Cython `cdef extern`) uses instead.  Each
 * entry cites the reference interface it replaces (paths relative to the
 * reference checkout).  All functions return 0 on success and a negative
 * SDP_E* code on failure; sdp_last_error() gives the message of the last
 * failure on the calling thread.  No C++ exceptions cross this boundary.
 * Host pointers are owned by the caller for the duration of a call; device
 * memory is owned by the library and tied to the handle that allocated it.
 */
#ifndef SDP_HIP_H
#define SDP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDP_OK          0
#define SDP_EINVAL     -1   /* bad argument (maps to ValueError / AssertionError) */
#define SDP_EDIM       -2   /* state dimension outside 1..4 (multilinear_cython.pyx:46-47 raises Exception) */
#define SDP_EHIP       -3   /* HIP runtime error (RuntimeError) */
#define SDP_ENOMEM     -4   /* device allocation failed (MemoryError) */
#define SDP_ECOMM      -5   /* RCCL error or librccl not loadable */
#define SDP_EMODULE    -6   /* model code object could not be loaded, or was not built for this problem */

#define SDP_F64 0
#define SDP_F32 1

/*
 * How a problem handle stores its per-node arrays ON THE DEVICE.  Host arrays
 * passed to / returned by this API (value, policy, index, per-node control
 * boxes) are ALWAYS in the reference's order: C order of the state grid, last
 * axis fastest (stodynprog.py:272,478); the library converts on the device.
 *   SDP_LAYOUT_NODES    device order = host order (generic kernels);
 *   SDP_LAYOUT_COLUMNS  axis 0 fastest (numpy.moveaxis(A, 0, -1)): the column
 *                       kernels for storage-separable models read axis-0
 *                       pencils coalesced.  node_begin/node_end and the parts of
 *                       sdp_problem_attach_comm are indices in DEVICE order and
 *                       must then be multiples of orders[0] (whole columns).
 */
#define SDP_LAYOUT_NODES   0
#define SDP_LAYOUT_COLUMNS 1
/* Sweep kernel of a SDP_LAYOUT_NODES handle:
 *   SDP_VARIANT_DIRECT  `sdp_sweep`: one global load per interpolation vertex;
 *   SDP_VARIANT_STAGED  `sdp_sweep_lds`: a workgroup owns a tile of nodes and
 *                       stages the value sub-block its next states reach in LDS
 *                       (csrc/sdp_staged_kernel.h).  Same results, bit for bit.
 * A SDP_VARIANT_DIRECT code object whose `sdp_meta` carries SDP_META_F_LEAD (several controlled
 * state variables next to an exogenous process, csrc/sdp_lead_kernel.h; lanes_per_node must be 1)
 * also exports `sdp_lead_reduce`: the library launches it over the whole grid before every `sdp_sweep`
 * (it needs the whole cost-to-go array on the device: not with attached parts) and owns the arrays it
 * fills.  Same results, bit for bit (reference stodynprog.py:639-691 with any `dims`, :57-81). */
#define SDP_VARIANT_DIRECT 0
#define SDP_VARIANT_STAGED 1

const char *sdp_last_error(void);

/* ---- device ------------------------------------------------------------- */
int sdp_device_count(int *count);
int sdp_set_device(int device);
/* name buffer >= 256 bytes; any out pointer may be NULL */
int sdp_device_info(int device, char *name, int *compute_units, int64_t *hbm_bytes,
                    char *gcn_arch /* >= 64 bytes */);
int sdp_synchronize(void);

/* ---- multilinear interpolation ------------------------------------------- */
/*
 * Replaces  multilinear_interpolation(smin, smax, orders, values, s)
 *   stodynprog/dolointerpolation/multilinear_cython.pyx:17-49
 *   (called from stodynprog/stodynprog.py:285 and dolointerpolation/multilinear.py:87)
 * smin,smax: [d]; orders: [d] int64 ('long[:]'); values: [n_v][S] C-contiguous
 * with S = prod(orders), last axis fastest; s: [d][n_s] C-contiguous;
 * out: [n_v][n_s].  All host pointers.  d must be 1..4 (else SDP_EDIM).
 * Uniform grid, linear extrapolation outside [smin,smax].
 */
int sdp_mlinterp_f64(int d, const double *smin, const double *smax, const int64_t *orders,
                     const double *values, int64_t n_v, const double *s, int64_t n_s,
                     double *out);
int sdp_mlinterp_f32(int d, const float *smin, const float *smax, const int64_t *orders,
                     const float *values, int64_t n_v, const float *s, int64_t n_s,
                     float *out);

/*
 * Interpolator object with device-resident values: replaces a
 * MlinInterpolator / MultilinearInterpolator instance that is set once and
 * evaluated many times (stodynprog.py:269-289, multilinear.py:83-91), e.g. the
 * policy look-ups of a simulation loop.  smin/smax are given as doubles (exact
 * for float32 grids); host_values: [n_v][S] of `dtype`; sdp_interp_eval takes
 * s: [d][n_s] and writes out: [n_v][n_s], both host arrays of `dtype`.
 */
typedef struct sdp_interp sdp_interp;
int sdp_interp_create(int dtype, int d, const double *smin, const double *smax,
                      const int64_t *orders, const void *host_values, int64_t n_v,
                      sdp_interp **out);
int sdp_interp_eval(sdp_interp *h, const void *host_s, int64_t n_s, void *host_out);
int sdp_interp_destroy(sdp_interp *h);

/* ---- value-iteration problem handle ---------------------------------------- */
typedef struct sdp_problem sdp_problem;
typedef struct sdp_comm sdp_comm;

/*
 * Discretised problem, i.e. what DPSolver holds after discretize_state /
 * discretize_perturb / control_steps (stodynprog.py:335-389, 432-463).
 */
typedef struct sdp_problem_desc {
    int32_t dtype;              /* SDP_F64 | SDP_F32: type of every real array below */
    int32_t d;                  /* state variables, 1..4 */
    int32_t nu;                 /* control variables, 1..4 */
    int32_t W;                  /* perturbation points; 0 = deterministic system */
    int64_t orders[4];          /* points per state axis (state_grid lengths) */
    const void *axes[4];        /* state_grid[k]: orders[k] reals (np.linspace values) */
    const void *wgrid;          /* perturb_grid[0]: W reals, NULL if W == 0 */
    const void *proba;          /* perturb_proba[0]: W reals */
    int32_t box_per_node;       /* 0: one box for all nodes, 1: arrays over nodes */
    int32_t lanes_per_node;     /* SDP_LANES the code object was built with */
    int32_t layout;             /* SDP_LAYOUT_NODES | SDP_LAYOUT_COLUMNS (see below) */
    int32_t variant;            /* SDP_VARIANT_DIRECT | SDP_VARIANT_STAGED (node layout only, see below) */
    const void *box_lo;         /* control_grids() lower ends: [nu] or [nu][S] reals */
    const void *box_hi;         /* upper ends */
    const int32_t *box_n;       /* points per control: [nu] or [nu][S] */
    int64_t node_begin;         /* slab of C-order node ids owned by this handle */
    int64_t node_end;           /*   ([0,S) on a single GPU) */
    const char *module_path;    /* gfx950 code object of the traced model (sdp_sweep, sdp_evalpol) */
    int32_t tile[4];            /* SDP_VARIANT_STAGED: node-tile shape the code object was built with */
    int32_t col_seg_nodes;      /* SDP_LAYOUT_COLUMNS with a row window (code object built with SDP_COL_ROWS
                                 * < orders[0]): nodes of a column one workgroup takes at most; 0 = no window */
    int32_t reserved;           /* must be 0 */
} sdp_problem_desc;

/* The code object must have been generated for THIS problem: it declares
 *     extern "C" __constant__ int32_t sdp_meta[SDP_META_WORDS]      (csrc/sdp_kernel_args.h)
 * -- real type, d, nu, perturbation or not, layout / kernel variant, the axis-0 length and the
 * perturbation count its column table is sized for, the control lattice of its control table --
 * and sdp_problem_create compares every field with `desc`.  A mismatch, or a code object
 * without `sdp_meta`, is refused with SDP_EMODULE and a message naming the field (the reference
 * raises on a bad shape too, multilinear_cython.pyx:46-47; it never returns stale values). */
int sdp_problem_create(const sdp_problem_desc *desc, sdp_problem **out);
int sdp_problem_destroy(sdp_problem *p);

/* J_next of value_iteration (stodynprog.py:466,498): S reals, C-order. */
int sdp_problem_set_value(sdp_problem *p, const void *host_V);
/* pol of eval_policy (stodynprog.py:693,723): [S][nu] reals (control values). */
int sdp_problem_set_policy(sdp_problem *p, const void *host_pol);

/* Lifted model constants.  A finite-horizon problem whose callables look data
 * up by time index (reference examples/01 Deterministic storage control/
 * det_storage_control.py:89, `P_req = p['P_req_data'][k]`, called from
 * DPSolver.bellman_recursion, stodynprog.py:582) is traced one time step at a
 * time; the constants of the step become the array `sdp_model_prm[n]` of the
 * code object (same `real` type as the problem), so all steps share one kernel.
 * Sets those n values for the launches that follow.  n must equal the count
 * the code object declares (0 when it declares none). */
int sdp_problem_set_params(sdp_problem *p, const void *values, int32_t n);

/*
 * One Bellman backup over the handle's node slab -- DPSolver.value_iteration
 * (stodynprog.py:466-534) fused with _value_at_state_vect (639-691) and the
 * interpolation (multilinear_cython.pyx:51-300).  Reads the value buffer,
 * writes J_k, the optimal control values and their flat lattice indices.
 * With a communicator attached the J_k slabs are all-gathered (RCCL) so that
 * every rank ends with the full J_k.  rel_dp != 0: J_ref = J_k[ref_index]
 * (ref_index: flat C-order index of the reference node, stodynprog.py:384);
 * J_k -= J_ref (stodynprog.py:523-525); *J_ref_out receives J_ref.
 * t_k: time index passed to the model of a non-stationary system.
 */
int sdp_problem_vi_sweep(sdp_problem *p, double t_k, int rel_dp, int64_t ref_index,
                         double *J_ref_out);

/*
 * n_iter fixed-policy backups -- DPSolver.eval_policy (stodynprog.py:693-775).
 * Starts from the value buffer, leaves the result in the J buffer.
 * J_ref_out: [n_iter] reference costs when rel_dp != 0 (may be NULL).
 */
int sdp_problem_eval_policy(sdp_problem *p, int32_t n_iter, int rel_dp, int64_t ref_index,
                            double *J_ref_out);

/*
 * The whole of one DPSolver.value_iteration call with host arrays in and out
 * (stodynprog.py:466-534; the reference takes J_next and returns fresh J_k, pol_k
 * every call): upload of host_V (NULL: keep the device's value buffer), backup,
 * relative-DP shift, download of J_k, of the policy values (host_pol, may be NULL)
 * and of the lattice indices (host_idx, may be NULL), queued back to back with one
 * synchronisation at the end.  Buffers from sdp_host_alloc (page-locked) move at
 * PCIe rate; any other host memory works too, slower.
 */
int sdp_problem_backup_host(sdp_problem *p, const void *host_V, double t_k, int rel_dp,
                            int64_t ref_index, void *host_J, void *host_pol, int32_t *host_idx,
                            double *J_ref_out);
/*
 * on != 0 (the default): on one GPU, grids of 8 MiB and more run the backup of
 * sdp_problem_backup_host in a few phases of the node range and send the finished rows of a
 * phase to the host under the kernel of the next one.  on == 0: one launch, then the downloads.
 * Same arrays either way.
 */
int sdp_problem_set_host_overlap(sdp_problem *p, int on);
int sdp_host_alloc(size_t bytes, void **out);      /* page-locked host memory */
int sdp_host_free(void *ptr);

/*
 * Batched closed-loop simulation without leaving the device -- the loop of the
 * reference's examples (examples/20 Searev storage control/storage_control.py:242-251:
 * `P_sto[k] = P_sto_law(E[k], Speed[k], Accel[k]); x[k+1] = sys.dyn(x[k], P_sto[k], w[k])`,
 * one interpolator call per step).  For B trajectories and T steps:
 *     u[k] = policy(x[k])            every control component interpolated from host_pol
 *                                    ([nu][S] control values on the state grid, C order;
 *                                    same arithmetic as sdp_mlinterp_*)
 *     x[k+1] = dyn(x[k], u[k], w[k]), g[k] = cost(x[k], u[k], w[k])    (the traced model)
 * host_x0 [d][B]; host_w [T][B] (NULL for a deterministic system); outputs host_x
 * [T+1][d][B] (x[0] = x0), host_u [T][nu][B], host_g [T][B] (may be NULL).  t0: time
 * index of step 0 for a non-stationary system.
 */
int sdp_problem_simulate(sdp_problem *p, const void *host_pol, int64_t B, int64_t T,
                         const void *host_x0, const void *host_w, double t0,
                         void *host_x, void *host_u, void *host_g);

/* Make the last J_k the next J_next without leaving the device. */
int sdp_problem_swap(sdp_problem *p);

int sdp_problem_get_value(sdp_problem *p, void *host_J);            /* S reals            */
int sdp_problem_get_policy(sdp_problem *p, void *host_pol /* [S][nu] reals or NULL */,
                           int32_t *host_idx /* [S] or NULL */);
/* With a communicator attached this is a collective call: the policy rows of the
 * other ranks' parts are all-gathered first (every rank must call it). */

/* HIP-event duration of the last sweep / eval kernel launch(es), milliseconds. */
int sdp_problem_last_kernel_ms(sdp_problem *p, double *ms);
/* Timed repetition for benchmarks: `reps` sweeps with ping-pong swap between
 * them; returns the total HIP-event time of the whole loop and of the sweep
 * kernels alone. */
int sdp_problem_bench_sweeps(sdp_problem *p, int32_t reps, int rel_dp, int64_t ref_index,
                             double *loop_ms, double *kernel_ms);

/* Diagnostic (tools/clock_probe.py): code objects built with -DSDP_STAMP=1 record
 * s_memtime / s_memrealtime of thread 0 of every workgroup at kernel entry and
 * exit.  enable != 0 allocates the stamp buffer (the launches that follow fill
 * it); host != NULL copies n_words 64-bit words out ([workgroup][4]); enable == 0
 * frees it.  Production code objects never touch the buffer. */
int sdp_problem_debug_stamps(sdp_problem *p, int enable, unsigned long long *host,
                             int64_t n_words);

/* ---- tabulated backup (models that cannot be traced into device code) ------- */
/*
 * Replaces the numeric part of DPSolver._value_at_state_vect
 * (stodynprog.py:677-690) for a batch of nodes whose dyn/cost callbacks were
 * evaluated on the host exactly as stodynprog.py:674,676 does:
 *   cell  = g[cell] + interp(V, x_next[:, cell])          (stodynprog.py:677)
 *   J[c]  = sum_w cell(c,w) * proba[w]  (W == 0: no expectation; 679-683)
 *   idx   = first-occurrence argmin over the node's controls (686)
 * Node n owns cells [cell_off[n], cell_off[n+1]), laid out [control][w].
 * The value array stays on the device between calls.
 */
typedef struct sdp_tab sdp_tab;
int sdp_tab_create(int d, const double *smin, const double *smax, const int64_t *orders,
                   const double *host_V, sdp_tab **out);
int sdp_tab_destroy(sdp_tab *t);
int sdp_tab_backup(sdp_tab *t, int64_t n_nodes, const int64_t *cell_off, int64_t W,
                   const double *proba, const double *x_next /* [d][n_cells] */,
                   const double *g /* [n_cells] */, double *J_out /* [n_nodes] */,
                   int64_t *idx_out /* [n_nodes] */);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ----------------------------
 * librccl.so is dlopen()ed on first use.  The product library binds librccl and nothing
 * else.  A TEST build of the same source (-DSDP_TEST_HOOKS, made by the test-suite into a
 * file of its own; sdp_test_hooks() returns 1 there, 0 in the product) additionally honours
 * the environment variable SDP_RCCL_LIBRARY: the nccl* entry points are then taken from
 * that shared object (tests/mock_rccl.cpp: a host-staged stand-in that lets several ranks
 * share the ONE GPU of the test box, which RCCL itself refuses).  sdp_comm_library()
 * reports the library the nccl* symbols came from. */
int sdp_test_hooks(void);
int sdp_comm_unique_id(char id[128]);                       /* rank 0 */
int sdp_comm_create(int rank, int nranks, const char id[128], sdp_comm **out);
int sdp_comm_destroy(sdp_comm *c);
const char *sdp_comm_library(void);      /* path/name the nccl* symbols came from, "" if not loaded */
/*
 * Shard the handle's backups over the ranks of `c`.  The node range (device
 * order) is cut into n_phases contiguous phases and every phase into one part
 * per rank: part_bounds[ph*(nranks+1) + r] .. [.. + r + 1] are rank r's nodes of
 * phase ph (whole columns in SDP_LAYOUT_COLUMNS).  Each backup then runs phase by
 * phase; the RCCL all-gather of a phase overlaps the kernel of the next one.
 * The handle must have been created with node range [0, S).
 */
int sdp_problem_attach_comm(sdp_problem *p, sdp_comm *c, int32_t n_phases,
                            const int64_t *part_bounds /* [n_phases][nranks+1] */);
/*
 * Alternative exchange for the sharded backups: every rank maps the other ranks' value / J
 * buffers (HIP IPC) and, phase by phase, WRITES its rows into them with device-to-device
 * copies on one stream per peer -- copy engines over xGMI, no compute units, all links of the
 * fully connected node at once -- instead of an RCCL all-gather per phase.  The writes are
 * one-sided, so two 1-word all-reduces bracket them: one before the first write of an API call
 * (every peer has returned from its previous call, i.e. finished reading its J), one at the end
 * of every backup (all rows have landed everywhere).  Same results.  Collective (all ranks
 * call it, after sdp_problem_attach_comm); on failure the handle keeps the RCCL exchange.
 */
int sdp_problem_enable_peer_exchange(sdp_problem *p);
/*
 * Unmaps the peers' buffers again (the RCCL exchange is back in place).  Local, but when problems are torn down
 * the order matters across the ranks: every rank unmaps, THEN (after a barrier of the caller's) the buffers are
 * freed by sdp_problem_destroy -- memory a peer process still maps must not be freed under it.
 */
int sdp_problem_disable_peer_exchange(sdp_problem *p);
/*
 * Sparse peer exchange (after sdp_problem_enable_peer_exchange): a backup sends a peer only the
 * rows of J that peer READS in its own backups -- for a column q of need_off, the sorted,
 * disjoint node ranges ranges[2k], ranges[2k+1], k in [need_off[q], need_off[q+1]) (whole columns
 * in the column layout; the relative-DP reference node must be in every list).  The host
 * computes them from the model: the cells the trailing next states of a rank's columns fall in.
 * A rank's J is then complete only there; sdp_problem_get_value completes it first (every rank
 * sends all its rows: collective), sdp_problem_backup_host sends all rows to begin with.
 * NULL, NULL switches back to sending everything.
 */
int sdp_problem_set_peer_needs(sdp_problem *p, const int64_t *need_off /* [nranks+1] */,
                               const int64_t *ranges /* [need_off[nranks]][2] */);
/* Sparse peer exchange: make the last backup's J complete on every rank (collective; a no-op
 * otherwise) -- what sdp_problem_get_value does before it downloads. */
int sdp_problem_complete_value(sdp_problem *p);
/*
 * Direct exchange (after sdp_problem_enable_peer_exchange; at most 8 ranks, the GPUs of one node):
 * the backup kernel that computes J[node] also STORES it into the mapped J buffer of every other rank
 * (with need lists set: of the ranks that read the node's column) -- stores over xGMI issued by the
 * kernel itself, spread over the whole sweep, instead of copies after each phase.  Nothing is left to
 * move when the kernel ends: a backup costs ONE launch and ONE 1-word all-reduce (the ranks meet
 * before anybody reads J or overwrites V).  Same results.  The reference has no counterpart (its only
 * parallel attempt is the commented-out Pool.imap over the nodes, stodynprog.py:503-509).
 */
int sdp_problem_set_direct_exchange(sdp_problem *p, int on);
/*
 * Send / receive exchange (an alternative to sdp_problem_enable_peer_exchange; call it BEFORE sdp_problem_set_peer_needs):
 * the sparse exchange through the collective library alone.  After each phase's kernel a rank ncclSends every other
 * rank the bounding range of the rows of that phase the other rank reads, and ncclRecvs its own -- one grouped
 * send / receive per pair of ranks and phase, no buffer of another process mapped, no store into one.  A rank's J is
 * complete where it reads; sdp_problem_complete_value / sdp_problem_get_value run the all-gather of every phase.
 * Without need lists it is the RCCL all-gather.  (The reference's counterpart: none; its loop over the nodes is
 * serial, stodynprog.py:511-515.)
 */
int sdp_problem_set_sendrecv_exchange(sdp_problem *p, int on);
/*
 * Reduced-array sweep (several controlled state variables, csrc/sdp_lead_kernel.h) on a sharded
 * problem: rows of the FIRST state axis the controls of a node reach on either side.  A rank then
 * reduces its own rows plus that many instead of the whole grid.  A guess is enough: a node whose
 * controls reach further is noticed by the kernel and evaluated from the value array itself (time,
 * not correctness).  rows < 0 (default): every rank reduces everything.
 */
int sdp_problem_set_lead_halo(sdp_problem *p, int64_t rows);
int sdp_comm_allreduce_max(sdp_comm *c, double *inout);    /* host scalar, for timing */
int sdp_comm_barrier(sdp_comm *c);

#ifdef __cplusplus
}
#endif
#endif /* SDP_HIP_H */
