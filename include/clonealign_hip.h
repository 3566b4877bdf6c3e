/*
 * clonealign_hip.h -- C ABI of the MI355X (gfx950) variational-inference engine that
 * replaces the TensorFlow-backed ELBO loop of kieranrcampbell/clonealign.
 *
 * Boundary: the reference's only compute call site is
 *   inference_tflow()            R/inference-tflow.R:71-481
 * whose graph build (:240-346) and session loop (:351-457) talk to TensorFlow through
 * reticulate.  Each entry point below replaces one `sess$run(...)` granularity of that
 * loop, so an R shim can either keep the R-level loop verbatim or hand the whole loop to
 * ca_run().  Plain C types only: no R, Python, torch or HIP types cross this boundary.
 *
 * Conventions
 *   - every function returns CA_OK (0) or an error code; ca_last_error() gives the text.
 *   - matrices are passed in ONE layout per problem (`layout`): CA_COL_MAJOR is what R
 *     hands over (column-major, i.e. Y is gene-major in memory), CA_ROW_MAJOR is C/numpy.
 *   - `eps` arguments are host pointers to S*G float32 standard normals, sample-major
 *     (eps[s*G + g]): the noise of ONE `qmu$sample()` evaluation
 *     (R/inference-tflow.R:268-269).  Passing the stream explicitly is what makes results
 *     reproducible across engines; NULL selects the built-in Philox4x32-10 stream keyed by
 *     ca_options.seed (draw index kept in the handle).
 *   - handles are independent and re-entrant (no globals); one host thread per handle.
 */
#ifndef CLONEALIGN_HIP_H
#define CLONEALIGN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_ABI_VERSION 6

typedef struct ca_engine* ca_handle;

enum ca_status {
  CA_OK = 0,
  CA_ERR_INVALID = 1, /* bad argument / shape (the reference's stopifnot()/stop() cases) */
  CA_ERR_HIP = 2,     /* HIP runtime failure */
  CA_ERR_NOMEM = 3,
  CA_ERR_NAN = 4,     /* "Initial elbo is NA" (R/inference-tflow.R:374-376) or NaN in the window test (:414) */
  CA_ERR_COMM = 5,    /* RCCL / peer-to-peer transport failure */
  CA_ERR_STATE = 6,
  CA_INTERRUPTED = 7  /* ca_run_ex(): the poll callback asked to stop; trace and variables are those of the last completed iteration */
};

enum ca_dtype { CA_F64 = 0, CA_F32 = 1, CA_I32 = 2, CA_U16 = 3, CA_U8 = 4 };
enum ca_layout { CA_ROW_MAJOR = 0, CA_COL_MAJOR = 1 };
enum ca_ystore { CA_YSTORE_AUTO = 0, CA_YSTORE_F32 = 1, CA_YSTORE_U16 = 2, CA_YSTORE_U8 = 3 };

/* Inputs of one fit: what inference_tflow() has in hand at R/inference-tflow.R:236,
 * after its gene filter (:117-124), saturate (:142-144) and initialisation (:204-235). */
typedef struct ca_problem {
  int64_t N;           /* cells held by THIS process (a shard when world > 1) */
  int32_t G, C;        /* genes, clones */
  int32_t K, P, S;     /* latent dims (:136), covariates (:147-153), MC samples (:268) */
  int32_t layout;      /* ca_layout, applies to Y, L, psi0, X, extra_loglik and ca_get_param outputs */
  int32_t y_dtype;     /* ca_dtype of Y */
  int32_t y_on_device; /* nonzero: Y is a device pointer on ca_options.device */
  const void* Y;       /* N x G counts (:190,355) */
  const double* L;     /* G x C copy number (:191) */
  const double* psi0;  /* N x K  (:204-208, `pcs`)   -- may be NULL when K == 0 */
  const double* loc0;  /* G      (:262, safe_inverse_softplus(mu_guess)); NULL = the data_init_mu = TRUE guess of :220-235
                          computed from the resident matrix.  Sharded (world > 1, ABI 6): every rank keeps its cells' part of the
                          per-gene sums and the guess is completed over ALL cells by the first reduction of the transport the
                          engine is given (ca_comm_init / ca_p2p_commit / ca_set_host_allreduce), before any pass can run */
  const double* X;     /* N x P covariates or NULL (:147-153) */
  const double* extra_loglik; /* N x C additive log-lik (allele term, :302-304) or NULL */
  /* Optional row / column selection of Y, so that the masks of ca_preprocess() (R/preprocess.R:141-147 returns filtered
   * COPIES) or the gene filter of R/inference-tflow.R:117-124 never cut the matrix on the host: Y is then the RAW
   * N_src x G_src matrix and cell n / gene g of the problem is its row cell_index[n] / column gene_index[g] (0-based,
   * strictly increasing).  NULL = identity (then N_src / G_src are ignored).  L, psi0, loc0, X, extra_loglik are always
   * given for the SELECTED cells and genes. */
  int64_t N_src;
  int32_t G_src;
  const int64_t* cell_index;  /* N entries or NULL */
  const int32_t* gene_index;  /* G entries or NULL */
  /* ABI 6: leading dimension of Y in ELEMENTS, 0 = dense.  CA_COL_MAJOR: distance between two columns (>= the rows of the source
   * matrix); CA_ROW_MAJOR: distance between two rows (>= its columns).  With it a block of ROWS of a column-major matrix -- what a
   * shard of R's N x G matrix is -- is handed over in place (Y = &M[lo], y_ld = nrow(M)), and only that block crosses PCIe. */
  int64_t y_ld;
} ca_problem;

/* Decomposition variants (bits of ca_options.variant_off: a set bit switches the variant OFF; 0 = the measured defaults).
 * They exist so that every fallback stays under test and every choice can be re-measured. */
enum ca_variant {
  CA_VAR_FUSED = 1 << 0,      /* fused two-eps sweep (monitor pass i + forward half of train pass i+1) */
  CA_VAR_FWD_MFMA = 1 << 1,   /* forward contraction on the matrix cores */
  CA_VAR_FWD_CELL = 1 << 2,   /* forward sweep and cell epilogue in one kernel */
  CA_VAR_BWD_MFMA = 1 << 3,   /* backward contraction on the matrix cores */
  CA_VAR_TAIL_FUSE = 1 << 4,  /* O(K + C) bodies as extra blocks of other launches */
  CA_VAR_ASYNC_Y = 1 << 5,    /* count-matrix products on the side stream */
  CA_VAR_PRE = 1 << 6,        /* next pass's per-gene prologue on the per-cell Adam kernel */
  CA_VAR_PAIR_ELBO = 1 << 7,  /* final ELBOs two draws per sweep */
  CA_VAR_PREP_FAST = 1 << 8,  /* wave-per-cell fit-constant kernel for u8 storage */
  CA_VAR_UPDATE_MERGE = 1 << 9, /* the update half of a train pass of the loop as ONE launch (k_update_merged: per-gene step + next prologue + int8 images, psi,
                                  q(z) logits, chi / alpha), the exponent bound taken by the next forward sweep; off: k_final_gene + k_adam_cell */
  CA_VAR_P2P_RIDE = 1 << 16,  /* sharded over the peer-to-peer transport: the backward sweep's column sums, the stream's finishing sums and the pending monitor
                                 pass's psi.(YW) sum ride in the sweep's and the all-reduce's launches (4 launches per iteration); off: k_yfinish + k_colsum launches */
  CA_VAR_RUN_GATE = 1 << 17,  /* ca_run: the update half of the next train pass is queued before the host has seen the ELBO its stop rule needs and waits ON THE
                                 DEVICE for the host's go / stop word (no launch latency between the decision and the update) -- for gate_timeout_us at most, then
                                 it gives up, stores nothing and is queued again after the decision; off: always queued after the decision */
  CA_VAR_S2_FUSE = 1 << 18,   /* mc_samples = 2: the monitor pass's two samples and the next train pass's two samples in ONE forward sweep (two operand sets,
                                 four draws on one exp per (cell, gene)); off: a sweep per pass */
  CA_VAR_FWD_BAL = 1 << 19,   /* small problems (one to six 16-cell tiles per CU): ONE eight-wave forward-sweep block per CU, the left-over tiles spread gene-wise over
                                 the blocks with their partial Z exchanged through tagged words (k_fwd_bal_ys, ca_fwdbal.hip.h): every SIMD holds two waves with equal
                                 work; off: 16- / 32-cell four-wave blocks, as many as the cells need */
  CA_VAR_BWD_TL3 = 1 << 20,   /* small problems (up to 18 432 cells): the matrix-core backward sweep takes three gene tiles per wave instead of four -- more and
                                 shorter wave jobs; off: four at every size */
  CA_VAR_SERIES = 1 << 21,    /* ABI 6: large problems with a rank-one exponent (K + P = 1, one MC sample, 3..8 clones, 1-byte storage; where G >= 2000 + 2.5e7 / N -- the round's first form: from 32k cells and 1.4e8
                                 counts): the loop's contraction in its SERIES form (ca_poly.hip, see CA_VARX_SERIES) -- moments over gene bins instead of the
                                 cells x genes sweeps; off: the matrix-core sweeps at every size */
  CA_VAR_P2P = 1 << 10,       /* one-shot peer-to-peer all-reduce (else ncclAllReduce) after ca_comm_init() */
  CA_VAR_FOLD_GSUM = 1 << 11, /* small problems: backward-sweep partials summed inside the per-gene kernel (else a k_colsum launch) */
  CA_VAR_Y_RIDE = 1 << 12,    /* the Y stream's blocks ride on the forward sweep's launch (else side stream / in line) */
  CA_VAR_Y_MFMA1 = 1 << 14,   /* both count-matrix products on the int8 matrix cores from ONE tiled copy of the 1-byte matrix, the column products
                                 through the transposing LDS read ds_read_b64_tr_b8 (ca_ymfma.hip.h; K = 1): the default since round 3, when its
                                 blocks ride on the forward sweep as long-lived stream blocks and its quantiser on the per-cell Adam kernel.
                                 Off: the vector stream (k_ypass / k_fwd_cell_mix_y) */
  CA_VAR_YFIN_RIDE = 1 << 15, /* the riding int8 stream's finishing sums (Y^T psi column sums, YW and the psi.(YW) partials) as extra blocks of the
                                 backward sweep's launch; the pending monitor pass's ELBO is then assembled one kernel later (per-gene kernel).
                                 Off: a finisher launch (k_yfinish) between the two sweeps */
  CA_VAR_RIDE_SEQ = 1 << 13   /* off: the riding stream as blocks of its own interleaved in the sweep's grid (k_fwd_cell_mix_y), never fused in
                                 sequence into the sweep's blocks (k_fwd_cell_seq_y; see CA_VARX_RIDE_SEQ) */
};
/* Opt-in variants (bits of ca_options.variant_on).  Those marked LAB were measured slower than what ships and are kept as the evidence for the choice (DESIGN_HISTORY.md):
 * since round 6 they are compiled into the lab library only (`make -C clonealign_amd/csrc lab`, -DCA_LAB); ca_create of the product library returns CA_ERR_INVALID for them. */
enum ca_variant_on {
  CA_VARX_Y_MFMA2 = 1 << 0,   /* LAB.  count-matrix products on the int8 matrix cores from TWO tiled copies (cell-tiled for Y.W,
                                 gene-tiled for Y^T.psi; ca_ymfma.hip.h): 6.0 TB/s per stream against 4.7 for k_ypass, but twice
                                 the bytes per iteration */
  CA_VARX_Y_MFMA1 = 1 << 2,   /* (round 2's opt-in for what is now the default, CA_VAR_Y_MFMA1; accepted and ignored) */
  CA_VARX_FOLD_ALWAYS = 1 << 3, /* backward-sweep partials summed inside the per-gene kernel at every size (default: up to 32k cells) */
  CA_VARX_RIDE_SEQ = 1 << 4,  /* LAB.  riding Y stream fused in sequence: every forward-sweep block also streams one unit of the count matrix,
                                 before or after its sweep (k_fwd_cell_seq_y), instead of separate stream blocks in the same grid */
  CA_VARX_P2P_SAME_DEVICE = 1 << 5, /* test rigs only: let ca_p2p_connect map a peer handle of the SAME process on the SAME device (refused
                                 otherwise: device-wide synchronising runtime calls of one handle would wait on the other's all-reduce) */
  CA_VARX_RUN_FWD = 1 << 6,   /* ca_run with the gated update: the forward sweep behind it is queued before the host's decision as well (its blocks return at
                                 their first instruction on "stop").  OPT-IN, and only for a process whose ONLY user of the GPU runtime is this engine's thread:
                                 the sweep is queued while the gated update may already be waiting for the host, and a runtime call made in that window can block
                                 behind another thread that holds a runtime lock while IT waits for the GPU (two engines of one process on one device did exactly
                                 that: the update then gives up after its 10 s and ca_run returns CA_ERR_STATE).  Worth about 1 us per iteration. */
  CA_VARX_BAL_TILES = 1 << 7, /* LAB.  balanced forward sweep of small problems (CA_VAR_FWD_BAL): a single-tile block of its own per left-over tile behind the sweep
                                 blocks, no exchange (the stream's blocks then go to the CUs without one); default: the left-over tiles cut gene-wise into chunks
                                 that the sweep blocks sweep beside their own tiles, partial Z exchanged through tagged words.  Level at few left-over tiles,
                                 slower at many */
  CA_VARX_SERIES = 1 << 8,    /* ABI 6: force the series form (CA_VAR_SERIES) at ANY size.  The form: where the exponent is rank one -- K + P = 1, one MC sample, 3..8
                                 clones: Z_nc = sum_g M_gc exp(x_n v_g) is one function of x per clone; genes binned by v, a 20-term expansion per bin (argument <= 2,
                                 float64): moments over genes, evaluation over cells, the same form on the way back.  No cells x genes sweep: O(N nb R C + G R C)
                                 instead of O(N G C) per pass; the cell epilogue is the sweep's.  Other shapes keep the matrix-core sweeps */
  CA_VARX_ASYNC_SMALL = 1 << 1 /* side stream also below 4e7 counts (small shards run the Y stream in line: the two cross-stream
                                 events cost more than the overlap returns there) */
};
#define CA_OPT_VERBOSE 0x80000000u /* in variant_off: print the decomposition picks to stderr */
/* decomposition parameters the heuristics pick; a non-zero ca_options.tune[id] overrides (CA_TUNE_FC_NBIG: -1 = one block size) */
enum ca_tune_id {
  CA_TUNE_GSPLIT = 0, CA_TUNE_FSPLIT = 1, CA_TUNE_FC_TL = 2, CA_TUNE_FC_NBIG = 3, CA_TUNE_CSPLIT = 4, CA_TUNE_CSPLIT_M = 5,
  CA_TUNE_TR = 6, CA_TUNE_RG = 7, CA_TUNE_COUNT = 8
};

typedef struct ca_options {
  double learning_rate;             /* :75,345 */
  double beta1, beta2, adam_eps;    /* tf.train.AdamOptimizer defaults 0.9, 0.999, 1e-8 */
  uint64_t seed;                    /* key of the built-in eps stream */
  int32_t device;                   /* HIP device ordinal */
  int32_t y_storage;                /* ca_ystore: on-device width of the count matrix */
  int32_t rank, world;              /* cell-sharded data parallel: shard `rank` of `world` */
  int32_t profile;                  /* bits 0..4: bitmask over ca_kernel_id, time those kernel classes with HIP events; bits 8..15: sampling
                                     * stride - 1 (0 = every launch; an event pair costs the stream 5-6 us, so live measurements sample) */
  uint32_t variant_off;             /* ca_variant bits to switch off (| CA_OPT_VERBOSE); 0 = defaults */
  int32_t tune[8];                  /* ca_tune_id overrides, 0 = heuristic */
  uint32_t variant_on;              /* ca_variant_on bits to switch on; 0 = defaults */
  int32_t ride_pattern;             /* dispatch order of the merged forward launch (sweep blocks + Y-stream blocks in one grid).  0 = default:
                                     * one LONG-LIVED stream block per CU leads the grid, each walks through every n-th unit of the count matrix
                                     * (from two units per CU up; below that the 2:1 interleave).  < 0: that many long-lived stream blocks.
                                     * > 0: (a << 8) | b = a sweep blocks, then b single-unit stream blocks, ...; periods that divide 8 put the
                                     * two kinds on disjoint XCDs.  Results do not depend on it (tests/test_gpu_parity.py) */
  int32_t comm_timeout_ms;          /* peer-to-peer all-reduce: how long a rank waits on the device for its peers' data before the call
                                     * gives up and the engine reports CA_ERR_COMM (0 = 10 000 ms) */
  int32_t gate_timeout_us;          /* ca_run with the gated update (CA_VAR_RUN_GATE): how long the queued update's relay block polls for the host's
                                     * go / stop word before the launch gives up -- it then stores nothing, the device goes idle, and the host queues
                                     * the update again after its decision (0 = 1000 us).  Not an error and not a result: only who waits for whom */
  int32_t reserved[2];              /* lab knobs of the series form: [0] cell blocks per CU (0 = 2), [1] = 1: its count-matrix stream on the side stream (measured slower) */
} ca_options;
/* (The library reads no tuning from the process environment.  Only the timing-lab build, -DCA_LAB -- never the product's .so, its
 *  ca_build_id() starts with "lab-" -- accepts the same switches as CA_* variables, and only when CLONEALIGN_DEBUG_ENV is set.) */

typedef struct ca_info {
  int64_t N;
  int32_t G, C, K, P, S;
  int32_t y_storage;         /* ca_ystore actually used */
  int32_t y_bytes_per_elem;
  int64_t y_device_bytes;    /* resident size of the count matrix */
  int64_t device_bytes;      /* total device allocation of this handle */
  int32_t gsplit, csplit;    /* gene / cell splits of the forward / backward sweeps */
  int32_t n_cu;
  int32_t fused_sweep;       /* 1: ca_run/ca_iterate fuse monitor pass i with the forward half of train pass i+1 */
  int32_t fwd_mfma;          /* 1: the fused sweep's forward contraction runs on the matrix cores (k_fwd_mfma) */
  int32_t bwd_mfma;          /* 1: the backward sweep's t = coef.L contraction runs on the matrix cores (k_bwd_mfma) */
  int32_t fsplit;            /* gene slices of the matrix-core forward sweep */
  int32_t fwd_cell;          /* 1: forward sweep and cell epilogue of the fused pass are ONE kernel (k_fwd_cell) */
  int32_t y_mfma;            /* the loop's count-matrix products on the int8 matrix cores: 0 no (k_ypass), 1 two tiled copies, 2 one */
  int32_t transport;         /* 0 none, 1 RCCL all-reduce, 2 host callback, 3 one-shot peer-to-peer (ca_transport) */
  int32_t y_ride;            /* 1: the Y stream's blocks ride on the fused forward sweep's launch (k_fwd_cell_mix_y): no launch of its own */
  int64_t red_n;             /* doubles all-reduced per train pass (= sharding.reduce_plan(...)["total"]) */
  /* ABI 4: which decomposition the shape selected (the tests assert that their shapes cross every threshold) */
  int32_t fwd_block_cells;   /* cells per block of the fused forward sweep: 16, 32 or 96 (0: no fused forward sweep) */
  int32_t fwd_blocks_big;    /* > 0: mixed launch, that many blocks of fwd_block_cells cells, the rest 32-cell blocks */
  int32_t fold_gsum;         /* 1: the backward sweep's per-gene partials are summed inside the per-gene kernel (unsharded small problems) */
  int32_t yfin_split;        /* 1: the Y stream's finishing step is split between the forward and backward launches */
  int32_t update_merge;      /* 1: the loop's update half is one launch (CA_VAR_UPDATE_MERGE) */
  int32_t fwd_balanced;      /* > 0: the fused forward sweep is the balanced small-problem form (CA_VAR_FWD_BAL), that many tiles per block */
  int32_t fwd_series;        /* ABI 6: 1: the loop's contraction takes its series form (CA_VAR_SERIES / CA_VARX_SERIES, ca_poly.hip) wherever the exponent range allows */
  int32_t reserved_;
  int64_t series_passes;     /* fused passes of this engine that ran in the series form ... */
  int64_t series_fallbacks;  /* ... and those the look ahead at the exponent range (max|psi| (max W - min W), plus what the Adam steps since can add) gave to the sweeps */
} ca_info;
enum ca_transport { CA_TRANSPORT_NONE = 0, CA_TRANSPORT_RCCL = 1, CA_TRANSPORT_HOST = 2, CA_TRANSPORT_P2P = 3 };

/* kernel classes reported by ca_get_kernel_times() */
enum ca_kernel_id {
  CA_KERNEL_FWD = 0,    /* Z = E.M sweep            (R/inference-tflow.R:278-292) */
  CA_KERNEL_BWD = 1,    /* reverse sweep of the same contraction (autodiff of :288-296) */
  CA_KERNEL_YPASS = 2,  /* Y.W and Y^T.psi stream   (the y*log p part of :294-296) */
  CA_KERNEL_CELL = 3,   /* per-cell epilogue: log-lik, softmax, ELBO partials (:294-308,332-333) */
  CA_KERNEL_OTHER = 4,  /* per-gene terms, reductions, Adam (:311-336,345-346) */
  CA_KERNEL_COUNT = 5
};

int ca_abi_version(void);
const char* ca_build_id(void); /* first 16 hex digits of the SHA-1 over the library's sources, as built */
/* Number of HIP devices this library's runtime sees (0 and CA_ERR_HIP when there is none).  Also initialises the runtime: a host
 * that loads a second HIP runtime into the process (PyTorch bundles its own) should call this right after loading the library,
 * so that the load order and the initialisation order agree (the runtime initialised second after being loaded first sees
 * "no ROCm-capable device"). */
int ca_device_count(int32_t* n);
int ca_default_options(ca_options* opts);

/* Build the engine: upload Y/L/init, precompute the fit constants (lgamma terms,
 * A = Y.log L, column sums), zero-initialise the variables as :240-272 does.
 * Replaces graph construction + `sess$run(init)` (:240-353). */
int ca_create(const ca_problem* problem, const ca_options* opts, ca_handle* out);
int ca_destroy(ca_handle h);
const char* ca_last_error(ca_handle h); /* h may be NULL: error of the last failed ca_create on this thread */
int ca_get_info(ca_handle h, ca_info* info);
int ca_synchronize(ca_handle h);
/* *busy = 1 while work queued on the engine's stream has not completed, else 0; never blocks, never changes anything (a poll hook can
 * watch the device drain while it holds the loop up). */
int ca_stream_busy(ca_handle h, int32_t* busy);

/* cell-sharded multi-GPU (one process per GPU): RCCL communicator over xGMI.
 * Rank 0 calls ca_comm_unique_id() and distributes the 128 bytes out of band. */
int ca_comm_unique_id(char id[128]);
int ca_comm_init(ca_handle h, const char id[128]);
/* One-shot peer-to-peer all-reduce over xGMI: the per-iteration payload is ~120 KB, i.e. latency-bound, so
 * instead of a ring every rank WRITES its summands into an inbox slab of every peer (IPC-mapped device memory) and adds the W
 * inboxes of its own slab in rank order 0..W-1 -- the same additions in the same order on every rank, so the replicas stay
 * bit-identical, in one kernel on the engine's stream, no host round trip.  Since round 4 every 8-byte store carries half a double
 * and the call's 32-bit tag: a summand is complete when it can be READ complete -- no fence, no flag, no counter (k_p2p_allreduce).  Setup: every rank exports a handle, the caller exchanges them out of band (like the RCCL id: torch.distributed /
 * MPI all-gather), every rank connects.  Needs peer access between the devices (same node); ranks may share a device. */
#define CA_P2P_HANDLE_BYTES 128
/* Setup is two-phase so that a one-sided failure cannot leave the other ranks waiting on the device:
 *   1. every rank: ca_p2p_export (allocates the slab in FINE-GRAINED device memory -- CA_ERR_COMM when that is unavailable, never
 *      a coarse-grained fallback -- and returns the handle); the caller all-gathers the handles out of band;
 *   2. every rank: ca_p2p_connect (maps the peers' slabs: IPC handles of other processes, the slab's own address for handles of
 *      the SAME process -- one host process may drive several devices, one handle per DEVICE on one host thread each; peer access
 *      is enabled between different devices; two handles of one process on one device are refused).  Launches nothing and waits
 *      for nobody;
 *   3. the caller agrees over its control plane whether step 2 succeeded on EVERY rank, then every rank calls
 *      ca_p2p_commit(h, all_ok): 1 makes the transport the engine's all-reduce and reduces the setup sums (the first call that
 *      waits for peers); 0 drops the mappings (next: ca_comm_init, or a host callback).
 * The device-side wait for the peers' data is bounded (ca_options.comm_timeout_ms): when it runs out the call gives up (its buffer
 * is then partly summed), every later all-reduce on this engine returns at once, and the next API call that synchronises returns CA_ERR_COMM.
 * The engine is then dead: destroy it (a fresh process is the recovery).  Calls that reduce are collective: every rank must make
 * the same sequence of ca_* calls, and a ca_run_ex poll hook must take the same decision on every rank. */
int ca_p2p_export(ca_handle h, char handle[CA_P2P_HANDLE_BYTES]);
int ca_p2p_connect(ca_handle h, const char* handles /* world x CA_P2P_HANDLE_BYTES, in rank order */);
int ca_p2p_commit(ca_handle h, int32_t all_ranks_ok);
/* us per all-reduce of n_doubles doubles on one device transport (CA_TRANSPORT_P2P or CA_TRANSPORT_RCCL) of this engine, n_calls
 * back to back on the engine's stream between two HIP events; collective.  bench.py reports it beside the iteration time. */
int ca_comm_benchmark(ca_handle h, int32_t transport, int32_t n_calls, int64_t n_doubles, double* us_per_call);
/* Known-answer test of the engine's ACTIVE all-reduce (whatever ca_info.transport says): n_rounds all-reduces whose summands are
 * (rank + 1) * pattern(i, round) + round / 2 -- every partial sum is an integer or a half, exact in a double in any order -- checked on the host
 * against W (W + 1) / 2 * pattern + W * round / 2.  Call sizes alternate as the loop's do -- n_doubles (the train pass's payload), min(n_doubles, 11)
 * (a monitor pass's), and a vector longer than the peer-to-peer inbox (several pieces per call) -- in bursts of eight back-to-back calls with
 * no host synchronisation between them (both inbox parities and the sequence tags at the loop's own pace); on the peer-to-peer transport
 * four calls of the RIDE form follow (slab fold + block-partial sum inside the all-reduce's launch, checked against the same sums made on the
 * host).  *n_bad = entries that came back wrong (0 = the transport adds what it should).  Collective.  The first time a transport runs on
 * hardware it has never run on (peer-to-peer across xGMI), this is what a launcher asks before it trusts it (bench.py: 48 rounds). */
int ca_comm_selftest(ca_handle h, int32_t n_rounds, int64_t n_doubles, int64_t* n_bad);
/* Alternative transport for world > 1 (MPI, gloo, tests): the engine hands the summand buffer to the
 * host callback, which must replace buf[0..n) by its sum over all ranks (same order on every rank)
 * and return 0.  Slower than RCCL (one device<->host round trip per reduction); same results. */
typedef int (*ca_host_allreduce_fn)(void* user, double* buf, int64_t n);
int ca_set_host_allreduce(ca_handle h, ca_host_allreduce_fn fn, void* user);

/* `sess$run(gamma_init)` + `sess$run(init_gamma)`   (:338-342,368-369) */
int ca_gamma_init(ca_handle h, const float* eps);
/* `sess$run(elbo)`                                   (:336,372,403,448) */
int ca_elbo(ca_handle h, const float* eps, double* elbo);
/* the three summands of :336 -- EE_p_y, E_log_p_p, E_log_q */
int ca_elbo_terms(ca_handle h, const float* eps, double terms[3]);
/* `sess$run(train)`: forward + backward + TF1 Adam on all variables (:345-346,401) */
int ca_step(ca_handle h, const float* eps);
/* gradients of the ELBO without the Adam update (for parity checks); fetch with ca_get_gradient */
int ca_gradients(ca_handle h, const float* eps, double* elbo);

/* The whole session loop :368-417: gamma init (draw 0), initial ELBO (draw 1), then up to
 * max_iter iterations of {train, monitor} with the 10-long mean |relative change| < rel_tol
 * stop rule.  eps_stream: n_draws*S*G floats consumed in order (needs >= 2 + 2*max_iter
 * draws) or NULL for the built-in stream.  elbo_trace receives 1 + iterations values. */
int ca_run(ca_handle h, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws,
           double* elbo_trace, int32_t* n_elbo);
/* The same loop with a host callback between iterations -- the reference's loop is R-level and can be interrupted or
 * observed every iteration (R/inference-tflow.R:394-417, progress bar :395-399).  poll(user, i, elbo_i) is called once
 * per ELBO value as the host learns it (i = 0 for the initial ELBO of :372, then 1, 2, ...), from the calling thread,
 * while the GPU already works on the backward sweep of the next train pass.  A non-zero return stops the loop exactly
 * like the convergence test does: the variables are those after iteration i, the trace has i + 1 values, and the call
 * returns CA_INTERRUPTED.  An R shim calls R_CheckUserInterrupt() through R_ToplevelExec() here. */
/* What a hook may do (ABI 5).  It may take as long as it likes -- a progress bar, a debugger, a sleep: the update the engine had queued
 * ahead of the hook's decision waits on the device for ca_options.gate_timeout_us at most, then gives up without storing anything and is
 * queued again after the hook returns; the fit is the same bit for bit, only that iteration loses the overlap.  It may call the read-only
 * entry points on this handle -- ca_get_param, ca_get_gradient, ca_get_info, ca_synchronize, ca_get_kernel_times, ca_stream_busy -- and sees
 * the variables after iteration `iter` (the first such call ends the queued update's wait the same way).  Every call that changes the
 * engine's state (ca_step, ca_set_param, ca_reinit, ca_run, ca_destroy, the ca_comm_* family ...) returns CA_ERR_STATE from inside a hook.
 * Sharded fits: the hook must take the same decision on every rank. */
typedef int (*ca_poll_fn)(void* user, int32_t iter, double elbo);
int ca_run_ex(ca_handle h, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws,
              double* elbo_trace, int32_t* n_elbo, ca_poll_fn poll, void* user);
/* n_iter iterations of {train, monitor} with no convergence test and no host sync inside
 * (the benchmark "step"); last_elbo may be NULL.  eps_stream: 2 n_iter draws consumed in order.  ABI 6: given ONE MORE draw (2 n_iter + 1), the
 * call's last sweep also makes the forward half of the first train pass of the NEXT ca_iterate call with it; that call picks it up when its own
 * first draw is that draw bit for bit (else it is dropped, nothing else changes): back-to-back calls then run n_iter sweeps each instead of
 * n_iter + 1.  The variables, the ELBOs and the number of draws CONSUMED (2 n_iter) are the same either way; the built-in stream (NULL) looks
 * one draw ahead by itself.  Any other call in between drops the carried half. */
int ca_iterate(ca_handle h, int32_t n_iter, const float* eps_stream, int64_t n_draws, double* last_elbo);
/* `replicate(20, sess$run(elbo))` (:447-454): values[n_rep], mean and sample sd */
int ca_final_elbo(ca_handle h, int32_t n_rep, const float* eps_stream, int64_t n_draws, double* values,
                  double* mean, double* sd);

/* psi initialisation on the device (R/inference-tflow.R:204-208): the first K principal components of the standardised
 * log2(Y + 1) matrix (prcomp(center = TRUE, scale = TRUE)), each scaled to unit variance (scale()), plus `noise`
 * (N x K in the problem's layout, the reference's rnorm(.., 0, 0.05), or NULL).  Blocked subspace iteration on the resident
 * count matrix (two passes over Y per iteration, K + 4 vectors); component signs are fixed by making the loading of largest
 * magnitude positive (prcomp's own signs are LAPACK's and arbitrary).  Overwrites psi; pcs_out (N x K) may be NULL. */
int ca_init_psi_pca(ca_handle h, const double* noise, int32_t n_iter, uint64_t seed, double* pcs_out);

/* Per-gene sums behind compute_correlations() (R/clonealign.R:318-334), one pass over the resident counts instead of
 * shipping Y back: clone_of_cell[n] in [0, C) or -1 ("unassigned", dropped at :319-320).
 *   T[g][c] = sum over cells assigned to clone c of y_ng      (G x C, problem layout)
 *   Syy[g]  = sum over assigned cells of y_ng^2
 * Pearson's r of (copy number of the assigned clone, counts) follows from T, Syy and the clone sizes on the host. */
int ca_clone_gene_sums(ca_handle h, const int32_t* clone_of_cell, double* T, double* Syy);

/* Fetch (:424-434).  name in {"mu","clone_probs","s","alpha","beta","psi","W","chi"} (the
 * reference's ml_params) or a raw variable {"loc","ls","gamma_logits","alpha_unconstr","v"}.
 * Output is float64 in the problem's layout; sizes: mu/loc/ls G, clone_probs/gamma_logits
 * N*C, s N, alpha/alpha_unconstr C, beta G*P, psi N*K, W G*K, chi/v K. */
int ca_get_param(ca_handle h, const char* name, double* out);
/* Overwrite a raw variable (same names as above, raw set + "psi","W","beta"); resets nothing else. */
int ca_set_param(ca_handle h, const char* name, const double* in);
/* after ca_gradients(): d ELBO / d variable, raw-variable names as in ca_set_param */
int ca_get_gradient(ca_handle h, const char* name, double* out);

/* A new restart on the same data: what run_clonealign()'s loop (R/clonealign.R:50-56) gets by calling inference_tflow() again --
 * all eight variables at their initial values (R/inference-tflow.R:240-273: W, v, beta, alpha_unconstr, ls, gamma_logits = 0,
 * psi = psi0, loc = loc0), fresh Adam state -- without uploading the count matrix or recomputing its fit constants.
 * psi0: N x K in the problem's layout (NULL when K = 0); loc0: G values or NULL for the loc ca_create() started from. */
int ca_reinit(ca_handle h, const double* psi0, const double* loc0);

/* accumulated HIP-event time per profiled kernel class since the last reset; ca_set_profile changes the mask */
int ca_get_kernel_times(ca_handle h, double ms[CA_KERNEL_COUNT], int64_t launches[CA_KERNEL_COUNT]);
int ca_reset_kernel_times(ca_handle h);
int ca_set_profile(ca_handle h, int32_t mask);

/* built-in eps stream (Philox4x32-10 + Box-Muller), host side: out[n] for draw `draw` */
int ca_eps_draw(uint64_t seed, uint64_t draw, int64_t n, float* out);

/* Allele-specific addend of the log-likelihood, needs no handle: replaces
 * construct_ai_likelihood() + beta_binomial_log_prob() (R/allele-specific.R:17-58) as evaluated at
 * R/inference-tflow.R:166-187 with alt = t(cov) - t(ref).  Host matrices in `layout` (ca_layout): clone_allele [V, C]
 * copy number of each clone at each variant, cov / ref [N, V] coverage and reference read counts per cell; out [N, C]
 * is what ca_problem.extra_loglik takes.  The term has no parameters, so it is computed once per fit.
 * err (optional, >= 256 bytes) receives the message on failure. */
int ca_allele_loglik(int64_t N, int32_t V, int32_t C, int32_t layout, const double* clone_allele, const double* cov,
                     const double* ref, int32_t device, double* out, char* err);

/* Gene / cell filters of preprocess_for_clonealign() (R/preprocess.R:93-147), needs no handle.
 * The two O(N G) statistics -- colSums(Y) and, after the gene filters, rowSums(Y[, kept]) -- are taken on the device
 * from the caller's raw matrix (any ca_dtype, either layout, host or device pointer); the O(G) decisions follow the
 * reference's order: copy number above max (:114-116), colSums <= min_counts_per_gene (:118-120), outlying gene means
 * (mean + nmads * mad, :59-63,123-128), equal copy number in all clones (:131-135), rowSums <= min_counts_per_cell
 * (:138-139).  Outputs: keep_gene [G], keep_cell [N] (1 = retained); gene_sums [G] / cell_sums [N] optional (may be NULL).
 * The caller subsets Y, L and the names with the masks (the reference returns the filtered matrices). */
typedef struct ca_preprocess_params {
  double min_counts_per_gene;         /* 20   */
  double min_counts_per_cell;         /* 100  */
  int32_t remove_outlying_genes;      /* 1    */
  int32_t remove_genes_same_copy_number; /* 1 */
  double nmads;                       /* 10   */
  double max_copy_number;             /* 6    */
} ca_preprocess_params;
int ca_preprocess(int64_t N, int32_t G, int32_t C, int32_t layout, int32_t y_dtype, int32_t y_on_device, const void* Y,
                  const double* L, const ca_preprocess_params* params, int32_t device, uint8_t* keep_gene,
                  uint8_t* keep_cell, double* gene_sums, double* cell_sums, char* err);

/* ------------------------------------------------------------------------------------------------------------------------------
 * ONE fit, cell-sharded over several devices of ONE process (ABI 6; SURVEY.md section 8b "multi-GPU via one process / 8 devices,
 * communicator created per fit", section 8e).  The reference's caller is a single R session: inference_tflow() is called once
 * (R/clonealign.R:262-280) and must come back with the whole fit.  A group holds one engine handle per entry of `devices`, rank r
 * on the cells cell_range(N, r, W) = [N r / W, N (r + 1) / W) of the problem, each driven by its own host thread (rank 0 by the
 * CALLING thread, so a poll hook runs where R's API may be used), joined by the first transport that passes the known-answer test
 * (ca_comm_selftest) on every rank:
 *     one-shot peer-to-peer by address (ca_p2p_*: the ranks' slabs are mapped by their own addresses, peer access between the
 *     devices)  ->  RCCL (ncclCommInitRank from the W threads)  ->  a host reduction between the threads (always available).
 * A transport that was committed and then failed its test leaves engines that cannot take another one: the group re-creates
 * them (second upload) and moves on -- ca_group_info says what happened and why.  The fit is the single-handle fit up to the
 * grouping of the fp64 cell sums; the ranks' replicated variables are bit-identical, and every call below returns what the
 * one-handle call of the same name returns, for ALL cells (cell-indexed outputs are gathered in the problem's layout).
 *
 * problem: as for ca_create, holding ALL cells (host pointers; a device pointer only when every rank is on that device);
 * loc0 = NULL and ca_group_init_psi_pca work sharded (sums completed over the transport).  opts: as for ca_create; device, rank and
 * world are ignored.  devices: n_devices HIP ordinals, rank order; an ordinal may repeat (test rigs on one GPU: host transport, or
 * peer-to-peer with CA_VARX_P2P_SAME_DEVICE).  transport: 0 = the chain above, or ONE ca_transport to insist on (error if it fails).
 * Calls on one group are made from one thread at a time, always the same one. */
typedef struct ca_group* ca_group_handle;
typedef struct ca_group_info {
  int32_t world;
  int32_t transport;          /* ca_transport in use */
  int32_t p2p_status;         /* 0 not tried, 1 in use, -1 set-up failed (no peer access, repeated device ...), -2 known-answer test failed */
  int32_t rccl_status;        /* same */
  int32_t rebuilds;           /* times the engines were created again after a committed transport failed */
  int32_t selftest_rounds;    /* all-reduces of the known-answer test the transport in use passed */
  int64_t N;                  /* all cells */
  char note[384];             /* why transports were skipped, in words */
} ca_group_info;
int ca_group_create(const ca_problem* problem, const ca_options* opts, const int32_t* devices, int32_t n_devices, int32_t transport,
                    ca_group_handle* out);
int ca_group_destroy(ca_group_handle g);
const char* ca_group_last_error(ca_group_handle g);   /* g may be NULL: the last failed ca_group_create on this thread */
int ca_group_get_info(ca_group_handle g, ca_group_info* info);
int ca_group_rank_handle(ca_group_handle g, int32_t rank, ca_handle* h);   /* read-only use (ca_get_info, ca_get_kernel_times ...) */
/* the one-handle calls, collectively on every rank; eps arguments are shared by the ranks (every rank must see the same draws) */
int ca_group_init_psi_pca(ca_group_handle g, const double* noise /* N x K, all cells, or NULL */, int32_t n_iter, uint64_t seed, double* pcs_out);
int ca_group_gamma_init(ca_group_handle g, const float* eps);
int ca_group_elbo(ca_group_handle g, const float* eps, double* elbo);
int ca_group_step(ca_group_handle g, const float* eps);
/* poll: called on the calling thread (rank 0's); its decision is handed to the other ranks, which wait for it at the same iteration */
int ca_group_run_ex(ca_group_handle g, int32_t max_iter, double rel_tol, const float* eps_stream, int64_t n_draws, double* elbo_trace,
                    int32_t* n_elbo, ca_poll_fn poll, void* user);
int ca_group_iterate(ca_group_handle g, int32_t n_iter, const float* eps_stream, int64_t n_draws, double* last_elbo);
int ca_group_final_elbo(ca_group_handle g, int32_t n_rep, const float* eps_stream, int64_t n_draws, double* values, double* mean, double* sd);
int ca_group_get_param(ca_group_handle g, const char* name, double* out);
int ca_group_reinit(ca_group_handle g, const double* psi0 /* N x K, all cells */, const double* loc0);
int ca_group_clone_gene_sums(ca_group_handle g, const int32_t* clone_of_cell /* N, all cells */, double* T, double* Syy);

#ifdef __cplusplus
}
#endif
#endif /* CLONEALIGN_HIP_H */
