/* spx_nnue_dev.h - development / test / measurement entry points of libspx_nnue.so.
 *
 * NOT part of the drop-in boundary (include/spx_nnue.h is): nothing here replaces a call of Stormphrax's src/eval. These are
 * the synthetic network presets every golden of this repository is pinned on (the reference's default net cannot be
 * fetched offline), generators of random legal positions and games for the harnesses, host emulations of the kernels'
 * per-lane code that let the CPU test suite check device logic without a GPU, the activations of the last call and what the
 * column-sliced pipeline's last walk held. They live in the same shared library so that tests and bench.py exercise exactly the
 * code the product runs; none of them launches a kernel the product path does not launch (the load-only gather probe of rounds
 * 2-4 and its 26 kernels left the library in round 5: experiments/r04_spx_probe.hip.txt). */
#ifndef SPX_NNUE_DEV_H
#define SPX_NNUE_DEV_H

#include "spx_nnue.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Repo-owned synthetic network (the default net cannot be fetched offline). preset: 0 tame, 1 wild (i32 wraps),
 * 2 extreme (i16 accumulator wraps too) - uniform random weights; 3 realistic: the weight SHAPE of a trained QA = 255 net
 * (src/eval/arch.h:36-50) - heavy-tailed piece-square rows of which about 40 % fit i8, 40 % have a handful of weights
 * beyond it and 20 % are densely wide, Laplace-like i8 threat and L1 weights - so that the measured rate does not rest on
 * every piece-square row being compact. Writes spx_synth_net_bytes() bytes. */
size_t spx_synth_net_bytes(void);
int spx_synth_net(uint64_t seed, int preset, void* buf, size_t nbytes);

/* Fault injection for the tests of the pipeline's fall-back paths: after spx_debug_enable_test_hooks(1) spx_ctx_set_option /
 * spx_ctx_create_opts accept ftx_fail_after K (the K-th scratch set of the column-sliced pipeline "does not fit"; -1: never) and
 * ftx_fail_launch K (the K-th pass of the pipeline from now on "fails to launch"; -1: never) - either way the batch and all later ones
 * are served by the one-kernel path. libspx_nnue.so does not export this function: there the two names are unknown options. */
int spx_debug_enable_test_hooks(int on);

/* Device-side intermediates of the last spx_eval_full* call on this context, for tests and profiling:
 * the u8 feature-transformer activations [n][1024] (stm half first; multilayer.h:92-152 activateFt output). */
int spx_debug_copy_ft(spx_ctx* ctx, size_t n, uint8_t* out);

/* Diagnostics of the column-sliced pipeline: start / end of each of the 256 workgroups of the last gather that used scratch set
 * `slot` (-1: the context's own, 0 .. 2: the pipelined calls' lanes), device clock ticks of 10 ns; out[512]. */
int spx_debug_ftx_block_times(spx_ctx* ctx, int slot, uint64_t* out);
/* ... and the plan those workgroups walked: out[0 .. 256) = plan words (CU slot c -> first segment at [c], number of segments at [32],
 * number of groups at [33], segments {king bucket, first group, end group} from word 64), then 1 280 + 17 words: the first sorted
 * position of every (king bucket, section lengths) bin and the buckets' starts. out[256 + 1 297]. */
int spx_debug_ftx_plan(spx_ctx* ctx, int slot, uint32_t* out);

/* `count` random legal positions (host chess core): game i plays min_ply .. max_ply uniformly random plies from the standard
 * start or, every dfrc_every-th game (0 = never), from a double-Chess960 start. Seeded and reproducible. */
int spx_random_positions(uint64_t seed, size_t count, int min_ply, int max_ply, int dfrc_every, spx_packed_pos* out);
/* The same kind of batch generated ON THE DEVICE (move generation + uniform move choice kernels, no evaluations) straight
 * into d_out (count records of device memory): game i plays min_ply + (draw mod range) random plies from the standard start
 * or, every dfrc_every-th game, a double-Chess960 start; a game that runs out of moves keeps its final position. Seeded and
 * reproducible, but a different stream of positions than spx_random_positions. The host only places the start pieces. */
int spx_random_positions_gpu(spx_ctx* ctx, uint64_t seed, size_t count, int min_ply, int max_ply, int dfrc_every, void* d_out);
/* One uniformly random legal move per record (datagen-style playouts in bulk): out[i] = positions[i] after the move,
 * moved[i] = 0 when the side to move has no legal move (out[i] = positions[i]). `moved` may be NULL. */
int spx_random_successors(uint64_t seed, const spx_packed_pos* positions, size_t n, spx_packed_pos* out, uint8_t* moved);

/* One random game as a viriformat stream (test / demo input; the scores are random). */
int spx_viri_random_game(uint64_t seed, int plies, int dfrc, void* buf, size_t capacity, size_t* nbytes);

/* perft of the host chess core (legal move generation check against published counts). */
uint64_t spx_perft(const char* fen, int depth);

/* Host emulation of the kernels' per-lane feature extraction (same SPX_HD code, run lane by lane on the CPU):
 * row ids of one perspective `colour` of `pos`. psq_rows capacity 32, threat_rows capacity 256. Test-only. */
int spx_debug_features(const spx_packed_pos* pos, int colour, uint32_t* psq_rows, int* n_psq, uint32_t* threat_rows,
                       int* n_threat);

/* Host emulation of the update kernel's DELTA derivation (same SPX_HD code, lane by lane): the rows perspective `colour`
 * loses (sub) and gains (add) between two boards one move apart - piece-square rows (capacity 8 each) and threat /
 * pawn-pair rows (capacity 288 each). *refresh = 1 (and empty lists) when the perspective is rebuilt instead: its king
 * changed bucket or mirror half (psq.h:264-283, nnue_state.h:118-128) or more than four squares differ. Test-only. */
int spx_debug_delta(const spx_packed_pos* parent, const spx_packed_pos* child, int colour, uint32_t* psq_sub,
                    int* n_psq_sub, uint32_t* psq_add, int* n_psq_add, uint32_t* threat_sub, int* n_threat_sub,
                    uint32_t* threat_add, int* n_threat_add, int* refresh);

/* Host evaluation of what SPX_ADJUST_WDL computes per position (same source as the kernel): Position::classicalMaterial
 * (src/position.h:515-521) of the record and wdl::normalizeScore (src/wdl.cpp:28-79) of `score` at it. Test-only. */
int spx_debug_wdl(const spx_packed_pos* pos, int32_t score, int32_t* material, int32_t* normalized);

/* Host evaluation of the datagen bookkeeping the device step kernel and the host self-play path share (same source):
 * counters[3] = the game's win / loss / draw ply counters (src/datagen/datagen.cpp:197-199), advanced by one searched move
 * with normalised white-point-of-view score `norm_score` at Position::plyFromStartpos `ply` (datagen.cpp:224-252);
 * *outcome = 0 / 1 / 2 (white loss / draw / win) or 255 = the game goes on. *insufficient = the material part of
 * Position::isDrawn for `pos` (src/position.cpp:639-666). Either output group may be skipped with NULL. Test-only. */
int spx_debug_datagen_rules(uint32_t* counters, int32_t norm_score, uint32_t ply, uint32_t* outcome,
                            const spx_packed_pos* pos, int* insufficient);

/* What the last packed walk of scratch set `slot` (-1 = the context's own, 0 .. 2 = a lane's) holds, 8 words: [0] groups of 8
 * perspectives, [1] stages, [2] steps of the cold sections and [3] of the LDS sections as the gather walks them PER COLUMN SLICE (a
 * step = 4 wave loads / LDS reads + 4 MFMAs; sections walk in pairs of steps), [4] cold threat / pawn-pair rows fetched through the
 * texture path, [5] rows read from LDS (piece-square + hot rows); the high-byte planes of wide piece-square rows are walked per slice
 * - an XCD drops the planes that are all zero in its slice -: [6] their steps as walked, SUMMED over the 8 slices, [7] plane slices
 * (128 B each) fetched, summed over the 8 slices. bench.py derives the gather's instruction counts from it. */
int spx_debug_ftx_walk(spx_ctx* ctx, int slot, uint32_t* out);

/* The row lists the column-sliced pipeline's extraction pass wrote for the LAST batch of scratch set `slot`, decoded back to the
 * net's row numbering (the reference's feature indices, psq.h:338-365 / nnue_state.cpp:309-354): for perspective 2 i + c of
 * position i (c = the perspective's colour, 1 = white) counts[3] = {piece-square rows, threat / pawn-pair rows, high-byte planes listed}
 * and rows[576] = the piece-square rows, then the threat / pawn-pair rows, then the piece-square rows whose high-byte plane is
 * fetched (rows behind the three counts are undefined). Order within a section is the extraction's, not the reference's:
 * compare as multisets. */
int spx_debug_ftx_lists(spx_ctx* ctx, int slot, size_t n_positions, uint32_t* counts, uint32_t* rows);

/* The hot set of the column-sliced gather, given instead of measured / read back (in slot order; *n = 0 before the first
 * calibration). spx_ctx_set_hot_rows takes up to 384 distinct row ids < 64 368 (n = 0: an empty set - every row through the
 * texture path) and waits for the context's streams; tests run the measured set, an empty one and an adversarial random one and
 * expect the same evaluations. */
int spx_ctx_set_hot_rows(spx_ctx* ctx, const uint32_t* rows, size_t n);
int spx_ctx_get_hot_rows(const spx_ctx* ctx, uint32_t* rows, size_t capacity, size_t* n);

#ifdef __cplusplus
}
#endif
#endif /* SPX_NNUE_DEV_H */
