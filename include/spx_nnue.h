/*
 * spx_nnue.h - C ABI of libspx_nnue: a batched, MI355X-native (gfx950) evaluator for Stormphrax's NNUE.
 *
 * Drop-in boundary. Stormphrax has no plugin/FFI layer; the seam is the C++ API of src/eval. Each entry point below
 * names the reference interface it replaces (paths relative to /root/reference). Signatures use only plain pointers
 * and sizes. Status: 0 = SPX_OK, otherwise an spx_status code; spx_last_error() returns the message of the calling
 * thread's last failure (the reference reports errors as `bool` + a line on stderr, src/eval/nnue.cpp:85-185,201-291).
 *
 * Encodings (identical to the reference's, src/core.h:336-350,389-415): squares a1 = 0 ... h8 = 63; colours
 * black = 0, white = 1; pieces type<<1|colour with pawn=0, knight=1, bishop=2, rook=3, queen=4, king=5.
 *
 * Threading: an spx_net is immutable and shareable; an spx_ctx belongs to one caller thread / one HIP stream at a
 * time (Lazy-SMP analogue: one NnueState per search thread, src/thread.h:147). Contexts on different GPUs are
 * independent. The library never frees or retains caller memory beyond a call (async variants: until the stream
 * reaches the enqueued work).
 */
#ifndef SPX_NNUE_H
#define SPX_NNUE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum spx_status {
    SPX_OK = 0,
    SPX_ERR_INVALID_ARG = 1,
    SPX_ERR_BAD_NET = 2,      /* header / size validation failed (nnue.cpp:85-185, loader.cpp:29-46) */
    SPX_ERR_HIP = 3,          /* a HIP runtime call failed */
    SPX_ERR_NO_DEVICE = 4,    /* no gfx950 device visible: the library never falls back to a CPU path */
    SPX_ERR_CAPACITY = 5,     /* batch larger than the context was created for */
    SPX_ERR_BAD_POSITION = 6  /* unparsable FEN / record without both kings */
} spx_status;

const char* spx_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Position record: marlinformat PackedBoard, 32 bytes (src/datagen/marlinformat.h:32-84).
 *   occupancy   bit i set = square i occupied
 *   pieces      one nibble per occupied square in ascending square order (low nibble first): type | colour<<3,
 *               colour bit set = BLACK, type 6 = rook that still has castling rights
 *   stm_ep      bit 7 = black to move; low 7 bits = en-passant square or 64
 * The evaluator reads occupancy, pieces and bit 7 of stm_ep only.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct spx_packed_pos {
    uint64_t occupancy;
    uint8_t pieces[16];
    uint8_t stm_ep;
    uint8_t halfmove;
    uint16_t fullmove;
    int16_t eval;
    uint8_t wdl;
    uint8_t extra;
} spx_packed_pos;

/* ------------------------------------------------------------------------------------------------------------------
 * Network (replaces eval::init / eval::getNetwork / eval::shutdown, src/eval/nnue.h:38-44, nnue.cpp:200-321).
 * `blob` is a CBNF file image: 64-byte header (src/eval/header.h:38-52) + arrays in the order of
 * preprocess/permute.cpp:33-56, in LOGICAL (unpermuted) column order - either plain, or with header flag 0x0001 and
 * the payload as one zstd frame (the form Stormphrax's release nets ship in, nnue.cpp:213-247; inflated through the
 * system's libzstd.so.1, loaded on demand). Validation mirrors nnue.cpp:85-185. The blob is copied; the caller may
 * free it after the call.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct spx_net spx_net;
int spx_net_load(const void* blob, size_t nbytes, spx_net** out);
void spx_net_free(spx_net* net);
const char* spx_net_name(const spx_net* net); /* eval::defaultNetworkName, nnue.h:44 */
/* FNV-1a 64 of the logical (uncompressed) payload: the same for a net loaded from its plain and from its zstd image */
uint64_t spx_net_digest(const spx_net* net);
/* How a context will serve this net's 11 264 piece-square rows (host-side, no device needed): rows whose 1 024 weights all fit
 * i8 (1 KiB copies), rows with at most 32 weights outside i8 (1 KiB copy + exact remainders in the full-refresh kernel) and
 * the rest (2 KiB i16 rows). See spx_ctx_compact_psq_rows / spx_ctx_near_psq_rows for what a given context actually did. */
int spx_net_psq_row_classes(const spx_net* net, uint32_t* fit_i8, uint32_t* near_compact, uint32_t* wide);

uint64_t spx_fnv1a64(const void* data, size_t nbytes);

/* ------------------------------------------------------------------------------------------------------------------
 * Device context: weights resident in HBM (re-laid out for the kernels), LUTs, scratch for `max_batch` positions.
 * Replaces the per-thread NnueState construction + setNetwork (src/eval/nnue_state.h:87-94, search.cpp:206,381).
 * Fails with SPX_ERR_NO_DEVICE when no HIP device is present.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct spx_ctx spx_ctx;
int spx_ctx_create(const spx_net* net, int device, size_t max_batch, spx_ctx** out);
/* As spx_ctx_create, with option flags. SPX_CTX_WIDE_PSQ_ROWS: every piece-square row is gathered from the 2 KiB i16
 * table, i.e. the lossless "compact row" optimisation (1 KiB u8 copies of rows whose weights all fit i8) is off - what a
 * net whose piece-square weights do not fit i8 gets anyway; bench.py reports this configuration next to the default. */
/* Full refreshes of 10 240 positions and more (pipelined calls: 6 144) take the column-sliced pipeline (stormphrax_amd/csrc/spx_ftx.hip: an extraction
 * pass writes every perspective's row lists, a counting sort groups them by king bucket and length, and the gather - XCD x
 * reads slice x of every row, the bucket's piece-square slab sits in LDS, rows are added up on the matrix pipe - runs in 0.7 x
 * the one-kernel path's time); smaller ones the one-kernel path (spx_ft_kernel). Results are bit-identical. The pipeline's
 * tables (89 MB per context) and scratch - 7.2 KB per position of a pass of min(max_batch, 65 536) positions: 4.6 KB of row lists
 * (2 x 576 words) and 2.6 KB of packed walks (the worst case of 10 one-KiB stages per group of 8 perspectives), i.e. ~470 MB per
 * scratch set at full size; stream-ordered calls use the context's set, pipelined calls one per lane (option eval_lanes, default 3):
 * ~1.9 GB in all - are allocated on the first such batch; if that fails the one-kernel path serves it.
 * SPX_CTX_ONE_KERNEL_FT (or option ftx = 0): never take the pipeline. SPX_CTX_SLICED_FT (or ftx = 1): take it (the default; kept
 * from round 4, when it was opt-in). SPX_CTX_WIDE_PSQ_ROWS implies SPX_CTX_ONE_KERNEL_FT. */
enum { SPX_CTX_WIDE_PSQ_ROWS = 1, SPX_CTX_SLICED_FT = 2, SPX_CTX_ONE_KERNEL_FT = 4 };
int spx_ctx_create_ex(const spx_net* net, int device, size_t max_batch, uint32_t flags, spx_ctx** out);
/* As spx_ctx_create_ex, with this context's own options: `options` = "name=value,name=value" (NULL / "" = none), the names of
 * spx_ctx_set_option below plus the three that shape what a context allocates (scratch_cap, compact_rows, near_rows). Applied after
 * SPX_OPTIONS from the environment (a later item overrides an earlier one). The thread-safe way to give creation-time options:
 * nothing process-global is touched. */
int spx_ctx_create_opts(const spx_net* net, int device, size_t max_batch, uint32_t flags, const char* options, spx_ctx** out);
/* Does a full refresh of n positions take the column-sliced pipeline on this context as things stand (enabled, n at or above the
 * threshold, no failed allocation of its tables so far)? Bit 0: a stream-ordered call (spx_eval_full[_device]) does; bit 1: a
 * pipelined call (spx_eval_full_device_async) does - its threshold is lower. 0 = the one-kernel path either way. */
int spx_ctx_sliced_ft(const spx_ctx* ctx, size_t n);
/* The pipeline's gather keeps the context's most popular threat / pawn-pair rows (option ftx_hot_rows, default 256 of 64 368) in
 * LDS beside the king bucket's piece-square slab: an LDS read costs half a trip through the CU's texture path, and on random legal
 * positions 256 rows serve 35 % of those fetches. The set is chosen from DATA: by default from the first batch that takes the
 * pipeline (one extra extraction pass + a histogram inside that call, for which the HOST WAITS - also in
 * spx_eval_full_device_async, once per context and after every change of ftx_hot_rows; a call on a stream that is being captured
 * into a hipGraph never calibrates - it runs with the set as it is -, and option ftx_auto_calibrate = 0 switches the automatic
 * choice off), or from the device-resident batch
 * handed to spx_ctx_calibrate (the first <= 65 536 positions of it; call it while no evaluation of this context is in flight - it
 * waits for the context's streams). Results never depend on the set (a row is added from wherever it lives); a self-play or data
 * rescoring host calibrates once on a sample of ITS positions. No reference counterpart: the reference has no cache to configure. */
int spx_ctx_calibrate(spx_ctx* ctx, const void* d_positions, size_t n);
/* Tuning knobs of a context - the analogue of the reference's tunable constants (src/tunable.h:161-169) and UCI options for this
 * path; none changes a result. spx_ctx_set_option changes one knob of one context between calls (not while a call of that context is
 * in flight); SPX_OPTIONS="name=value,name=value" in the environment - the only environment variable the library reads besides
 * LOCAL_WORLD_SIZE (self-play's host-thread share per rank) - applies to every context the process creates. Unknown names / malformed values: SPX_ERR_INVALID_ARG.
 *   ftx 0|1                 big full refreshes through the column-sliced pipeline (as SPX_CTX_SLICED_FT / _ONE_KERNEL_FT)
 *   ftx_min N               smallest batch that takes it (default 10 240; 6 144 for pipelined calls)
 *   ftx_hot_rows N          threat / pawn-pair rows the gather keeps in LDS beside the piece-square slab (default 256, at most 384)
 *   ftx_auto_calibrate 0|1  1 (default): the first big batch chooses that set (the host waits once inside that call); 0: only
 *                           spx_ctx_calibrate / spx_ctx_set_hot_rows do - until then the set is empty
 *   ftx_fold_sort 0|1       1 (default): a one-pass batch of that pipeline orders the positions for the MLP (by output bucket,
 *                           output.h:44-55) in the pipeline's own counting sort; 0: two sort launches of their own, as for every other batch
 *   pace_events 0|1         1 (default): a pipelined call records an event (nobody reads it) at the five points where a profile
 *                           would - with those records in the lanes' streams the pipeline settles into a 12 % faster steady state
 *   eval_lanes 2|3          scratch sets spx_eval_full_device_async rotates its batches over (default 3: the preparation of two
 *                           batches runs beside a gather)
 *   tiny_batch_max N        batches up to N positions skip the sorts (default 8 192)
 *   mlp_share_max N         positions up to which four waves share one MLP tile (default 8 192)
 *   ft_team_max N           full refreshes of up to N perspectives run one workgroup per perspective (default 512)
 *   update_chain_max N      fused updates of up to N records take the single-launch chain kernel (default 1 024)
 *   update_split_max N      updates of up to N records give every perspective a wave of its own (default 16 384)
 *   stream_acc_min N        updates from N records on access the arena non-temporally (default 32 768)
 *   refresh_waves N         waves of the rebuild pass behind an update (default 0 = automatic)
 *   ft_blocks_per_cu N, update_blocks_per_cu N    grid caps of the one-kernel full refresh / the update kernel (48 / 24)
 *   king_sort 0|1           one-kernel path: perspectives in king-bucket order (default 1)
 *   replay_paths -1|0|1, replay_segment N         spx_acc_replay_tree: by heavy paths / by levels / its own choice; plies per path segment (default 8)
 *   selfplay_graph 0|1, selfplay_graph_plies N, selfplay_trace 0|1    spx_selfplay_run: plies captured into hipGraphs (default) or
 *                           launched one by one; plies per graph (0 = automatic; even, 2..16); a timing line on stderr at the end
 * At creation only - spx_ctx_create_opts or SPX_OPTIONS - (they shape what a context allocates): scratch_cap N (below),
 * compact_rows 0|1, near_rows 0|1 (A/B of the lossless 1 KiB copies of piece-square rows that fit i8 / almost fit i8).
 * (The fault-injection hooks of the fall-back tests - ftx_fail_after, ftx_fail_launch - are unknown options here: they exist only
 * behind spx_debug_enable_test_hooks of include/spx_nnue_dev.h, which this library does not export.) */
int spx_ctx_set_option(spx_ctx* ctx, const char* name, int64_t value);
/* Positions the context keeps intermediates for at once: min(max_batch, scratch_cap = 4 Mi by default). The
 * spx_eval_full* entry points accept up to max_batch positions per call and walk them in chunks of this size (an
 * HBM-filling batch costs 36 bytes per resident position: record in, score out); the arena entry points (spx_acc_*) and
 * spx_movegen take at most this many records per call. */
size_t spx_ctx_scratch_batch(const spx_ctx* ctx);
void spx_ctx_destroy(spx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-device group: one context per GPU inside ONE process, for a native host that embeds the library (the
 * reference builds one NnueState per search thread, src/search.cpp:206,381 - here one context per device). The weights
 * are uploaded once per member; batches are cut into contiguous shards (sizes differ by at most one, spx_group_shard) and
 * every member evaluates its shard on its own host thread - no collective on the data path, results identical to one
 * context evaluating the whole batch. `devices` = HIP device ordinals (NULL / 0 = every visible device; an ordinal may
 * repeat: two members on one GPU). spx_group_member exposes a member for every per-context entry point above / below.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct spx_group spx_group;
int spx_device_count(int* count); /* visible HIP devices; SPX_ERR_NO_DEVICE (and *count = 0) when there is none */
int spx_group_create(const spx_net* net, const int* devices, size_t n_devices, size_t max_batch_per_device,
                     uint32_t flags, spx_group** out);
void spx_group_destroy(spx_group* group);
size_t spx_group_size(const spx_group* group);
spx_ctx* spx_group_member(spx_group* group, size_t index);
int spx_group_shard(const spx_group* group, size_t n, size_t index, size_t* lo, size_t* hi);
/* == spx_eval_full / spx_adjust over the whole group; n may be up to spx_group_size x max_batch_per_device */
int spx_group_eval_full(spx_group* group, const spx_packed_pos* positions, size_t n, int32_t* out);

/* ------------------------------------------------------------------------------------------------------------------
 * Full-refresh evaluation == n x NnueState::evaluateOnce(pos, pos.stm()) (src/eval/nnue_state.cpp:612-634): raw
 * network output in centipawn-like units from the side to move's point of view, BEFORE eval::adjustStatic's
 * contempt/clamp (src/eval/eval.cpp:25-28).
 *   spx_eval_full        host buffers in/out, synchronous (H2D, kernels, D2H on the context's stream); any n
 *                        (processed in chunks of max_batch)
 *   spx_eval_full_device device-resident buffers, enqueued on `stream` (a hipStream_t; NULL = the context's own
 *                        stream) without synchronising; n <= max_batch. Two exceptions to "without synchronising": the first
 *                        batch that takes the column-sliced pipeline allocates its tables, and - unless ftx_auto_calibrate = 0 or
 *                        a set was given - chooses the gather's hot rows, for which the host waits once (spx_ctx_calibrate).
 *                        Under hipStreamBeginCapture: make one call of the same size outside the capture first (allocations);
 *                        a captured call never calibrates; capture an EVEN number of calls per graph (the output-bucket sort
 *                        alternates two histogram buffers, each call clearing the other one's - as spx_selfplay_run's graphs do).
 * ---------------------------------------------------------------------------------------------------------------- */
int spx_eval_full(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, int32_t* out);
int spx_eval_full_device(spx_ctx* ctx, const void* d_positions, size_t n, void* d_out, void* stream);

/* Pipelined variant for streams of batches (rescoring, throughput runs): the call returns at once and consecutive
 * calls overlap - the context alternates two internal streams and scratch sets so that the sorts and the MLP of one
 * batch run beside the feature-transformer kernel of the next (the FT kernels themselves are chained). The inputs must
 * be valid when the call is made and stay untouched, like d_out, until the batch is done: *done_event (a hipEvent_t,
 * owned by the context, valid until two more async calls) or spx_ctx_synchronize(ctx). Results are bit-identical to
 * spx_eval_full_device. Allocates the lanes' scratch sets on first use (~1.1 KB per position of max_batch per lane, plus the
 * column-sliced pipeline's 7.2 KB per position of a pass where it runs - see spx_ctx_create_ex). */
int spx_eval_full_device_async(spx_ctx* ctx, const void* d_positions, size_t n, void* d_out, void** done_event);
/* waits for everything the context has enqueued on its own streams */
int spx_ctx_synchronize(spx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Incremental path: an arena of accumulator "slots" resident in HBM (4 KiB of i16 accumulators + the 32-byte record
 * per slot) replaces NnueState's accumulator stack (src/eval/nnue_state.h:47-83,87-116). A caller that used to do
 *     state.reset(pos)                          -> spx_acc_refresh(slot of the root, pos)
 *     pos2 = pos.applyMove(m, state.push())     -> record (parent slot, child slot, packed pos2); no observer needed
 *     state.evaluate(pos2, stm)                 -> spx_acc_update(records of this ply...) then spx_acc_eval(slots)
 *     state.pop()                               -> nothing: the parent's slot is still materialised
 * Batching rule: the records of ONE spx_acc_update call are independent parent->child pairs - every parent slot is
 * already materialised and no child slot of the call is a parent in the same call (process a tree level by level).
 *   spx_acc_refresh  == NnueState::reset (nnue_state.cpp:539-560) for n (position, slot) pairs
 *   spx_acc_update   == updatePsq + applyThreatUpdates / refreshes (nnue_state.cpp:34-87,356-394,458-536): the delta is
 *                       derived on the device from the parent slot's record and the child record
 *   spx_acc_eval     == evaluateNetwork on materialised slots (nnue_state.cpp:396-438); stm comes from the slot's record
 * Host-buffer variants synchronise; *_device variants take device pointers and enqueue on `stream` (NULL = context's).
 * A context is single-threaded and STREAM-ORDERED: the *_device calls of one context share its scratch (sort buffers,
 * refresh counters, activations), so calls on different streams must be ordered against each other by the caller (events);
 * the *_async entry points do that ordering themselves on the context's two internal lanes.
 * ---------------------------------------------------------------------------------------------------------------- */
/* spx_acc_reserve sizes the arena (never shrinks). Growing it keeps every materialised slot: accumulators and records are
 * copied into the new allocation (both arenas exist for the duration of the call). */
int spx_acc_reserve(spx_ctx* ctx, size_t n_slots);
int spx_acc_refresh(spx_ctx* ctx, const spx_packed_pos* positions, const uint32_t* slots, size_t n);
int spx_acc_update(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                   const spx_packed_pos* child_positions, size_t n);
int spx_acc_eval(spx_ctx* ctx, const uint32_t* slots, size_t n, int32_t* out);
int spx_acc_refresh_device(spx_ctx* ctx, const void* d_positions, const void* d_slots, size_t n, void* stream);
int spx_acc_update_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                          const void* d_child_positions, size_t n, void* stream);
int spx_acc_eval_device(spx_ctx* ctx, const void* d_slots, size_t n, void* d_out, void* stream);
/* update immediately followed by evaluation of the children (== push + applyMove + evaluate, the common search step):
 * one launch fewer and no re-read of the fresh accumulators.
 * child_slots == NULL (every spx_acc_update_eval* entry point): EVAL-ONLY children. The children's accumulators exist in
 * registers only, their activations go straight to the MLP and nothing is written to the arena - what the reference does
 * for a node it evaluates and immediately unmakes (NnueState::evaluate on the top of the stack, nnue_state.cpp:598-610:
 * no accumulator outlives the pop). A depth-1 search evaluates ~35 siblings per parent this way and then materialises
 * only the move it plays (one ordinary spx_acc_update): 97 % of the arena writes of the materialising form disappear. */
int spx_acc_update_eval(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                        const spx_packed_pos* child_positions, size_t n, int32_t* out);
int spx_acc_update_eval_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                               const void* d_child_positions, size_t n, void* d_out, void* stream);
/* NnueState::ensureUpToDate (nnue_state.cpp:636-697) over a pending PATH, then evaluate (:598-610): ply 0's parent is the
 * materialised `parent_slot`, ply k's parent is ply k - 1; every ply's accumulators are written to child_slots[k] (the
 * reference leaves every stack entry on the path clean) and *out (may be NULL) receives the evaluation of the last
 * position. One kernel launch whatever the length - the accumulator travels in registers from ply to ply - instead of
 * one synchronous call per pending ply. Host buffers, synchronous; n <= 8192. */
int spx_acc_update_chain_eval(spx_ctx* ctx, uint32_t parent_slot, const uint32_t* child_slots,
                              const spx_packed_pos* child_positions, size_t n, int32_t* out);

/* A recorded make/unmake TREE (BASELINE config 3: the PUSH / POP / EVAL stream of a search, src/thread.cpp:46-67,
 * src/thread.h:116-122) replayed natively: node 0 is the root (NnueState::reset), node k > 0 was reached from node
 * parents[k] < k by one move and positions[k] is its record. Every node gets arena slot k (the arena is grown to n_nodes
 * slots); the tree is processed LEVEL BY LEVEL - all updates of one depth are one batch of independent records - or, when
 * it is deep and narrow (depth >= 32, on average <= 4 096 nodes per level: a real alpha-beta search), by HEAVY PATHS - every
 * path one wavefront pair that carries the accumulator from ply to ply, all paths whose head's parent exists one launch -
 * on buffers uploaded once, with no host synchronisation in between; then the n_evals nodes of eval_nodes are
 * evaluated (NnueState::evaluate at those nodes) into out[]. *gpu_ms (optional) = device time from the root refresh to the
 * last evaluation. The reference walks the same tree depth-first with one lazily updated accumulator stack
 * (nnue_state.cpp:636-697); the values are identical. */
int spx_acc_replay_tree(spx_ctx* ctx, const spx_packed_pos* positions, const uint32_t* parents, size_t n_nodes,
                        const uint32_t* eval_nodes, size_t n_evals, int32_t* out, double* gpu_ms);

/* Pipelined variant for chains of plies (self-play style loops, trace replays level by level): returns at once; the
 * update kernels of consecutive calls run in call order (a ply's parents may be the previous call's children), while the
 * sort and the MLP of one call overlap the update kernel of the next on the context's second internal stream. Inputs and
 * d_out must stay untouched until *done_event (a hipEvent_t owned by the context, valid until two more async calls) or
 * spx_ctx_synchronize(ctx). Results are bit-identical to spx_acc_update_eval_device. */
int spx_acc_update_eval_device_async(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                     const void* d_child_positions, size_t n, void* d_out, void** done_event);

/* Page-locked host memory for batch buffers (positions, slots, scores). Optional: every host-buffer entry point accepts
 * ordinary memory, but copies from/to page-locked buffers run as plain DMA at PCIe speed instead of being staged by the
 * runtime (the self-play driver builds its child batches in such buffers). Returns NULL on failure. */
void* spx_host_alloc(size_t nbytes);
void spx_host_free(void* ptr);

/* ---- post-processing of raw evals on the device -------------------------------------------
 * eval::adjustStatic (src/eval/eval.cpp:24-27: + contempt[stm], clamp to +-24999) and eval::adjustEval
 * (src/eval/eval.cpp:30-67: material scaling + optimism, halfmove damping, optional correction / 2048, clamp) applied
 * in place to n raw evals that belong to the n records (piece counts, side to move and the halfmove clock are read from
 * the 32-byte records). spx_eval_full + SPX_ADJUST_STATIC == eval::staticEvalOnce; + SPX_ADJUST_EVAL ==
 * eval::adjustedStaticEval. The correction-history table itself stays with the search on the host: pass its
 * per-position `correction(pos, keyHistory)` values, or NULL for adjustEval<false>. i32 arithmetic wraps where the
 * reference's would overflow (undefined there). */
enum {
    SPX_ADJUST_STATIC = 1,
    SPX_ADJUST_EVAL = 2,
    /* datagen's view of a score (Searcher::runDatagenSearch, src/search.cpp:237-238), applied after the stages above:
     * WHITE_POV negates the evals of positions with black to move; WDL maps the score through wdl::normalizeScore
     * (src/wdl.cpp:28-79: f64 cubic in Position::classicalMaterial of the record, std::round; zero and decisive scores
     * pass through) - the value the reference's adjudication counters compare (src/datagen/datagen.cpp:224-252).
     * Parity target of the f64 part: the cubic is evaluated with FUSED multiply-adds, as the reference's clang x86-64 release
     * builds contract it (-ffp-contract=on is that compiler's default); the CPU oracle is built with -ffp-contract=off. The
     * two flavours were scanned over every material value x every score in +-26 000 (4.16 M pairs): the rounded integers
     * never differ, so a GCC / MSVC build of the reference that does not contract is the same parity target (tolerance: none). */
    SPX_ADJUST_WHITE_POV = 4,
    SPX_ADJUST_WDL = 8
};
typedef struct spx_adjust_params {
    int32_t contempt[2];             /* eval::Contempt, by colour: [0] black, [1] white (eval.h:31) */
    int32_t optimism[2];             /* eval::Optimism (eval.h:32) */
    int32_t scaling_value[5];        /* pawn, knight, bishop, rook, queen (tunable.h:161-165) */
    int32_t material_scaling_base;   /* tunable.h:167 */
    int32_t optimism_base;           /* tunable.h:168 */
    int32_t optimism_material_scale; /* tunable.h:169 */
    uint32_t stages;                 /* SPX_ADJUST_STATIC | SPX_ADJUST_EVAL */
} spx_adjust_params;
void spx_adjust_defaults(spx_adjust_params* params); /* the reference's default tunables, no contempt / optimism, both stages */
int spx_adjust(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, const spx_adjust_params* params,
               const int32_t* corrections, int32_t* evals);
int spx_adjust_device(spx_ctx* ctx, const void* d_positions, size_t n, const spx_adjust_params* params,
                      const void* d_corrections, void* d_evals, void* stream);
int spx_group_adjust(spx_group* group, const spx_packed_pos* positions, size_t n, const spx_adjust_params* params,
                     const int32_t* corrections, int32_t* evals);

/* As spx_acc_update_eval_device, but the number of records is read on the DEVICE from *d_count (u32, at most `capacity`;
 * `capacity` sizes the launches and must fit the context): a producer kernel - spx_movegen_device's d_total - feeds the
 * update without a host round trip. Records beyond *d_count are not touched. */
int spx_acc_update_eval_device_counted(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                       const void* d_child_positions, const void* d_count, size_t capacity, void* d_out,
                                       void* stream);
/* Per-kernel GPU timing of subsequent spx_eval_full* calls (HIP events recorded on the stream the kernels run on,
 * at most max_calls calls). spx_profile_end waits for the last recorded call and returns the summed durations of the
 * sort kernels, the feature-transformer kernel and the MLP kernel in milliseconds. Used by bench.py's roofline line. */
int spx_profile_begin(spx_ctx* ctx, size_t max_calls);
int spx_profile_end(spx_ctx* ctx, double* sort_ms, double* ft_ms, double* mlp_ms, size_t* calls);
/* ... and, of the calls the last spx_profile_end summed up, the time between the sorts and the feature-transformer stage's main
 * kernel (ft_ms: spx_ft_kernel, or the column-sliced pipeline's gather): on stream-ordered calls the pipeline's preparation
 * kernels (extraction, counting sort, plan); on pipelined calls what was left of the wait for the other lane's stage. */
int spx_profile_last_prepare_ms(const spx_ctx* ctx, double* prepare_ms);

/* Active feature rows of a batch, both perspectives summed (what a full refresh gathers): algorithmic bytes =
 * 2048 * psq_rows + 1024 * threat_rows (+ 36 B per position of record and score). Host-side count. */
int spx_count_rows(const spx_packed_pos* positions, size_t n, uint64_t* psq_rows, uint64_t* threat_rows);

/* The same count split by what THIS context's kernels fetch: piece-square rows served from the 2 KiB i16 table
 * (psq_wide_rows), from their 1 KiB u8 copy (psq_compact_rows), and the 1 KiB threat / pawn-pair rows. Requested bytes
 * of a full refresh = 2048 * wide + 1024 * (compact + threat). */
int spx_ctx_count_rows(const spx_ctx* ctx, const spx_packed_pos* positions, size_t n, uint64_t* psq_wide_rows,
                       uint64_t* psq_compact_rows, uint64_t* threat_rows);

/* Number of piece-square rows (of 11264) whose weights all fit i8: the context keeps a 1 KiB u8 copy of those and the
 * full-refresh kernel fetches it instead of the 2 KiB i16 row (identical sums, fewer bytes). Net dependent; 0 when
 * the context was created with option compact_rows=0. Reported by bench.py next to the algorithmic byte count. */
uint32_t spx_ctx_compact_psq_rows(const spx_ctx* ctx);
/* Number of NEAR-compact piece-square rows: all but at most 32 of the 1 024 weights fit i8. The full-refresh kernel fetches
 * their 1 KiB copy too (those weights clamped) and adds the exact remainders from a side table - identical sums again;
 * the incremental kernels read such a row from the i16 table. 0 with option compact_rows=0 or near_rows=0. */
uint32_t spx_ctx_near_psq_rows(const spx_ctx* ctx);

/* ------------------------------------------------------------------------------------------------------------------
 * Host helpers (position plumbing for harnesses; counterparts: src/position.cpp FEN parsing, marlinformat pack,
 * src/datagen/datagen.cpp:146-171 random openings).
 * ---------------------------------------------------------------------------------------------------------------- */
int spx_pos_from_fen(const char* fen, spx_packed_pos* out);
int spx_pos_to_fen(const spx_packed_pos* pos, char* buf, size_t nbytes);
int spx_pos_to_mailbox(const spx_packed_pos* pos, uint8_t mailbox[64], int* stm);
/* Position::applyMove for a move in UCI notation (castling as king-takes-rook, e.g. e1h1, or standard e1g1);
 * legality is checked against the generated legal moves (src/position.cpp:109-197, Position::moveFromUci). */
int spx_pos_apply_uci(const spx_packed_pos* pos, const char* uci, spx_packed_pos* out);
/* The record of every node of a recorded tree - or FOREST - of moves (trace replays): node k is a root when its move is
 * empty (it takes the next record of `roots`), otherwise positions[parents[k]] (parents[k] < k) after `moves + 6 k`
 * (UCI text, NUL padded to 6 bytes; "0000" = a null move: same board, other side to move, no en-passant square). */
int spx_tree_expand_uci(const spx_packed_pos* roots, size_t n_roots, const uint32_t* parents, const char* moves, size_t n,
                        spx_packed_pos* out);
/* Position::applyMove WITH the reference's observer: besides the successor record, the UpdateContext the BoardObserver
 * would have captured (src/eval/nnue_state.h:28-31,118-186; src/eval/nnue.cpp:490-599): piece-square subs/adds in event
 * order, threat descriptors added/removed (x-ray extensions/retractions included, cancelling pairs kept exactly as the
 * reference emits them), per-colour refresh flags, pawn bitboards before/after, kings after. The GPU update path does not
 * need it (it derives deltas from the two boards); it exists for hosts that keep the reference's bookkeeping and as a
 * parity check of that bookkeeping (tests/golden/deltas.txt). */
typedef struct spx_threat_desc {
    uint8_t attacker, attacker_sq, attacked, attacked_sq; /* ThreatDescriptor, psq.h:30-35 */
} spx_threat_desc;
typedef struct spx_move_delta {
    uint8_t n_sub, n_add, n_threats_added, n_threats_removed;
    uint8_t sub_piece[2], sub_sq[2], add_piece[2], add_sq[2];
    uint8_t psq_refresh[2], threat_refresh[2]; /* [black, white] */
    uint8_t kings[2];
    uint8_t reserved[2];
    uint64_t pawns_before[2], pawns_after[2];
    spx_threat_desc threats_added[128], threats_removed[128];
} spx_move_delta;
int spx_pos_apply_uci_observed(const spx_packed_pos* pos, const char* uci, spx_packed_pos* out, spx_move_delta* delta);
/* Incremental update from observer-captured deltas - the reference's own bookkeeping applied on the device
 * (updatePsq nnue_state.cpp:34-87, applyThreatUpdates :356-394 with generatePpRows :163-307, refreshes :458-536).
 * Same batching rule and same results as spx_acc_update; `out` (host) / `d_out` (device) may be NULL, otherwise the
 * children are evaluated as in spx_acc_update_eval. */
int spx_acc_update_observed(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                            const spx_packed_pos* child_positions, const spx_move_delta* deltas, size_t n,
                            int32_t* out);
int spx_acc_update_observed_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                   const void* d_child_positions, const void* d_deltas, size_t n, void* d_out,
                                   void* stream);
/* viriformat game streams (src/datagen/viriformat.cpp:28-63: PackedBoard + {u16 move, i16 score}* + 4 zero bytes per
 * game) -> one record per played move (the position BEFORE the move, `eval` = the recorded score, `wdl` = the game's
 * outcome). `unfiltered` (optional, one byte per record) tells which of them the reference's marlinformat output keeps:
 * Marlinformat::push drops a position when the side to move is in check or the played move is noisy - a capture, an
 * en passant or a queen promotion (datagen.cpp:254, position.cpp:683-689, marlinformat.cpp:31-36): 1 = stored, 0 =
 * filtered. Pass out = NULL to count only. */
int spx_viri_expand(const void* data, size_t nbytes, spx_packed_pos* out, int16_t* scores, uint8_t* unfiltered,
                    size_t capacity, size_t* n_positions, size_t* n_games);
/* The same expansion on the device (one thread per game replays the moves on the packed records): byte-identical output
 * for well-formed streams, two to three orders of magnitude faster than the host replay, which validates every move
 * against the legal-move generator - this one trusts the stream as the reference's own reader does; *bad_games counts
 * games that hit a move whose from-square holds no piece of the side to move (the rest of such a game repeats the
 * last position). `bad_games` may be NULL. */
int spx_viri_expand_gpu(spx_ctx* ctx, const void* data, size_t nbytes, spx_packed_pos* out, uint8_t* unfiltered,
                        size_t capacity, size_t* n_positions, size_t* n_games, size_t* bad_games);
/* datagen's two other output formats (datagen.cpp:340-346), converted from a viriformat stream - what the reference would
 * have written for the same games had it been started with "marlinformat" / "fen":
 *   spx_viri_to_marlinformat: the unfiltered positions as PackedBoard records with eval = the recorded score and wdl = the
 *     game's outcome, back to back (Marlinformat::push / writeAllWithOutcome, marlinformat.cpp:32-57);
 *   spx_viri_to_fen: one text line per unfiltered position, "<fen> | <score> | <0.0 / 0.5 / 1.0>" + '\n' (fen.cpp:32-66).
 * out = NULL counts only (*n_records / *n_bytes); SPX_ERR_CAPACITY when `capacity` (records / bytes) is too small.
 * The reference's two filters are both applied: per position (in check / noisy move, datagen.cpp:254) and per game - the move
 * that ends a game through Position::isDrawn is pushed as filtered whatever it is (datagen.cpp:264-268); viriformat does not
 * record that flag, so the converters look at the position after a game's last move themselves (50-move clock, repetition
 * against the game's own positions, insufficient material). Tablebase adjudication (datagen.cpp:270-281) does not exist here. */
int spx_viri_to_marlinformat(const void* data, size_t nbytes, spx_packed_pos* out, size_t capacity, size_t* n_records,
                             size_t* n_games);
int spx_viri_to_fen(const void* data, size_t nbytes, char* out, size_t capacity, size_t* n_bytes, size_t* n_games);

/* ---- legal move generation + make-move on the device (takes the host out of the self-play loop) ----
 * For each of the n records: every legal move (viriformat move word, src/datagen/viriformat.cpp:37-52; castling as
 * "king takes rook") and the 32-byte record of the position after it, byte-identical to what the host chess core
 * produces (spx_pos_legal_moves below). Children of position i are children[first[i] .. first[i] + count[i]); blocks of
 * different positions are placed in no particular order. parents[k] = parent_values[i] of the source position (or i
 * when parent_values is NULL) - pass the positions' accumulator slots to feed spx_acc_update_eval_device directly.
 * in_check[i] tells mate from stalemate when count[i] == 0. *total receives the number of children generated; if it
 * exceeds `capacity` nothing beyond capacity was written and the call returns SPX_ERR_CAPACITY.
 * The _device variant is asynchronous: d_total is a device u32 the caller reads back after the stream. */
int spx_movegen(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, const uint32_t* parent_values,
                spx_packed_pos* children, uint16_t* moves, uint32_t* parents, uint32_t* first, uint32_t* count,
                uint8_t* in_check, size_t capacity, size_t* total);
int spx_movegen_device(spx_ctx* ctx, const void* d_positions, size_t n, const void* d_parent_values, void* d_children,
                       void* d_moves, void* d_parents, void* d_first, void* d_count, void* d_in_check, size_t capacity,
                       void* d_total, void* stream);
/* Host chess core, one position: legal moves (<= 256) in viriformat encoding and, if `children` is not NULL, the
 * records after them; *in_check = side to move is in check. The parity reference of spx_movegen. */
int spx_pos_legal_moves(const spx_packed_pos* pos, uint16_t* moves, spx_packed_pos* children, int* n, int* in_check);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched self-play driver (BASELINE config 4 shape; the game loop of src/datagen/datagen.cpp:96-318): n_games concurrent
 * games, per ply every legal move of every game is evaluated in one incremental batch (score = -staticEval(child), i.e. a
 * depth-1 "search" - Stormphrax's alpha-beta search is out of scope), viriformat game records appended to out_path (NULL =
 * discard). The reference's rules, restated once in spx_device_math.h and pinned by tests/_datagen_rules.py:
 *   random 8-9 ply openings (datagen.cpp:146-171); the opening VERIFICATION filter (:176-190: a first search whose
 *   normalised score exceeds +-500 discards the opening - here the first ply's own search); scores from WHITE's point of
 *   view (search.cpp:237), leaves clamped like eval::adjustStatic; the win / loss / draw adjudication counters on the
 *   WDL-normalised score at the material and plyFromStartpos of the position searched (search.cpp:238,
 *   datagen.cpp:224-252); Position::isDrawn of the new position (50-move rule unless checkmate, threefold repetition,
 *   insufficient material; position.cpp:603-667) overriding the adjudication and recording score 0 (datagen.cpp:264-268);
 *   recorded score 0 when |score| <= 2 (:283); checkmate / stalemate (:213-221). Plus this driver's own ply cap (draw).
 * Default: the games LIVE ON THE GPU - legal moves and child records from spx_movegen, the ~35 siblings per position
 * evaluated WITHOUT being stored (eval-only children), one step kernel per ply doing search, bookkeeping, repetition keys,
 * game records and the seating of new games from a device-side opening pool, one materialising update for the moves
 * played; the two halves of the seats run on the context's two lanes, each half's per-ply launch chain captured once as a
 * hipGraph (two plies per graph launch, four on runs of >= 8 games per seat; option selfplay_graph=0 = direct
 * launches, option selfplay_graph_plies = plies per graph), and the host reads ~100
 * bytes of counters plus the finished games per ply. SPX_SELFPLAY_HOST_MOVEGEN selects the host chess core for moves, openings and bookkeeping (the
 * same rules; the validation path). Needs a context whose max_batch holds a ply's children of half the seats (48 * n_games
 * is always enough); reserves 2 * n_games + 1 arena slots (129 * n_games with host move generation, 9 * n_games + 1 with a search).
 * Multi-GPU: games are independent - run one process per GPU with its own seed / slice of games.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct spx_selfplay_params {
    uint32_t n_games;        /* concurrent games (e.g. 4096) */
    uint32_t target_games;   /* games to finish in total */
    uint32_t max_plies;      /* draw by length */
    uint32_t opening_plies;  /* random opening plies before play starts (0 = 8, plus a coin flip as datagen.cpp:153) */
    uint32_t dfrc;           /* 1 = double-Chess960 starts */
    int32_t temperature_cp;  /* pick uniformly among moves within this margin of the best (0 = greedy) */
    uint32_t host_threads;   /* host worker threads of the HOST move generation path; 0 = auto: min(16, usable CPUs = cgroup quota /
                              * LOCAL_WORLD_SIZE). The device-resident games need one host thread */
    uint32_t flags;          /* 0 = moves generated on the device; SPX_SELFPLAY_HOST_MOVEGEN = host chess core instead;
                              * | SPX_SELFPLAY_SEARCH_NODES(k): a live fixed-node search picks the moves (device path only) */
    uint64_t seed;
} spx_selfplay_params;
enum { SPX_SELFPLAY_HOST_MOVEGEN = 1 };
/* Live fixed-node search in place of the depth-1 policy (datagen's Searcher::runDatagenSearch with its soft node limit,
 * search.cpp:212-239, datagen.cpp:78-80): every game runs its own iterative-deepening alpha-beta, ONE node expanded per game
 * and round, the node's children evaluated in the same batch as every other game's - k = nodes a search may expand before it
 * stops at the end of an iteration (1 .. 2^24 - 1; k = 1 plays exactly the depth-1 games, temperature included; k > 1 plays
 * the search's best move and records its score; a decisive score ends the game, datagen.cpp:224-226). The rules are with
 * SearchStepParams in csrc/spx_kernels.h and restated in tests/_search_rules.py. 7 more arena slots per game;
 * stats.evals = leaves evaluated, stats.steps = nodes expanded. */
#define SPX_SELFPLAY_SEARCH_NODES(k) ((uint32_t)(k) << 8)
typedef struct spx_selfplay_stats {
    uint64_t games, positions, evals, steps;
    uint64_t outcomes[3];    /* white loss / draw / white win (datagen/common.h:24-28) */
    double seconds, gpu_seconds;
} spx_selfplay_stats;
int spx_selfplay_run(spx_ctx* ctx, const spx_selfplay_params* params, const char* out_path, spx_selfplay_stats* stats);
/* The same over a device group (BASELINE configs[3]: "games sharded 2/4/8 MI355X") from ONE native process: member r plays
 * its contiguous share of n_games / target_games on its own device and host thread (seed + r, output <out_path>.<r>.vf);
 * no exchange step; stats are summed (seconds: the slowest member). Members need spx_selfplay_run's context capacity for
 * their share. */
int spx_group_selfplay_run(spx_group* group, const spx_selfplay_params* params, const char* out_path, spx_selfplay_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* SPX_NNUE_H */
