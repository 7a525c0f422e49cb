// Kernel parameter blocks and launch wrappers (spx_kernels.hip). Device pointers only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

// u8 row table byte order within a dword (host relayout in spx_api.cpp and the kernels' widening must agree): columns
// (c, c + 2, c + 1, c + 3) - even bytes widen with v_and_b32, odd bytes with v_perm_b32.

namespace spx {

constexpr int kHistOut = 256;      // where a sort-histogram buffer holds the 8 output-bucket counts the MLP maps its tiles from
constexpr int kHistWords = 1024;  // one sort-histogram buffer (counts + cursors for up to 256 first keys and 8 output keys)

constexpr size_t kAccSlotBytes = 2 * 1024 * 2;  // one arena slot: 2 perspectives x i16[1024] (psq + threat combined)

// device-resident network tables shared by the feature-transformer and update kernels
struct FtTables {
    const int16_t* psqW;     // [11264][1024] i16, logical column order
    const uint8_t* thrW;     // [64368][1024] u8: value+128, columns interleaved per lane (see relayout in spx_api)
    const int16_t* ftBias;   // [1024]
    const uint32_t* lut;     // kLutWords threat LUT
    const uint64_t* deltaTab;  // kDeltaTabWords ray / knight masks + pseudo-attack sets (threat-delta derivation)
    const uint32_t* outlierTab;  // [kPsqRows][kOutlierCap] remainders of the near-compact rows, or nullptr (net has none)
};

struct FtParams {
    const void* positions;   // spx_packed_pos[nPositions] (32 B records)
    uint32_t nPositions;
    const uint32_t* order;   // optional permutation of perspective ids (2*pos + colour), or nullptr
    FtTables t;
    uint8_t* ftOut;          // mode A: [nPositions][1024] u8 activations (stm half, then nstm half)
    uint8_t* accOut;         // mode B (ftOut == nullptr): accumulator arena ...
    const uint32_t* slots;   //         ... slot of each position   (both outputs may be set)
    const uint32_t* nPerspPtr;  // optional: number of entries of `order` lives on the device (deferred refresh list)
    uint32_t* clearWord;        // optional: a device word this launch zeroes (the refresh counter of the NEXT update)
    uint8_t* slotRecords;    //         ... and the per-slot record store [nSlots][32]
};

struct UpdateParams {
    uint32_t nRecords;
    const uint32_t* nRecordsPtr;   // optional: the actual record count lives on the device (<= nRecords, which then
                                   // only sizes the grid) - lets a producer kernel feed this one without a host sync
    const uint32_t* parentSlots;   // [nRecords] materialised slots
    const uint32_t* childSlots;    // [nRecords] slots to write (distinct from every parent of this batch); nullptr (with
                                   // ftOut set) = eval-only children: nothing is stored in the arena
    const void* childPositions;    // spx_packed_pos[nRecords]: the boards after the move
    FtTables t;
    uint8_t* arena;                // [nSlots][kAccSlotBytes]
    uint8_t* slotRecords;          // [nSlots][32]
    const uint8_t* deltas;         // spx_update_observed_kernel only: spx_move_delta[nRecords] (1080 B each)
    uint8_t* ftOut;                // optional fused evaluation: [nRecords][1024] activations of the children ...
    uint8_t* stagedRecords;        // ... and [nRecords][32] their records (input of the MLP's bucket sort)
    uint32_t* refreshList;         // spx_update_kernel: ids (2 * record + colour) of the perspectives to rebuild ...
    uint32_t* refreshCount;        // ... and their number (zero on entry); the FT kernel launched next consumes both
};

struct ChainParams {               // spx_update_chain_kernel: whole pending PATHS (NnueState::ensureUpToDate) in one launch
    uint32_t nChains;
    const uint32_t* parentSlots;   // [nChains] the materialised slot each path starts from
    const uint32_t* first;         // [nChains] index of the path's first ply in the arrays below; nullptr = unit paths (path i = ply i)
    const uint32_t* count;         // [nChains] plies of the path (>= 1): ply k's parent is ply k - 1 (unused for unit paths)
    const uint32_t* childSlots;    // [total plies] slot every ply's accumulators are written to; nullptr = eval-only
    const void* childPositions;    // spx_packed_pos[total plies]
    FtTables t;
    uint8_t* arena;
    uint8_t* slotRecords;
    uint8_t* ftOut;                // optional: [nChains][1024] activations of each path's LAST position ...
    uint8_t* stagedRecords;        // ... and [nChains][32] its record
};

struct SlotActParams {
    uint32_t nSlots;
    const uint32_t* slots;
    const uint8_t* arena;
    const uint8_t* slotRecords;
    uint8_t* ftOut;           // [nSlots][1024]
    uint8_t* stagedRecords;   // [nSlots][32] contiguous copy of the slots' records (input of the bucket sort)
};

struct AdjustParams {        // spx_adjust_params flattened (include/spx_nnue.h)
    int32_t contempt[2];
    int32_t optimism[2];
    int32_t scalingValue[5];
    int32_t materialScalingBase, optimismBase, optimismMaterialScale;
    uint32_t stages;
    uint32_t nPositions;
    const uint64_t* positions;    // records as u64[4]
    const int32_t* corrections;   // nullable
    int32_t* evals;               // in place
};

struct MovegenParams {              // spx_movegen_kernel (spx_movegen.hip)
    const uint64_t* positions;      // records as u64[4]
    uint32_t nPositions;
    const uint32_t* parentValues;   // nullable: value recorded for every child of position i (e.g. its accumulator
                                    // slot); default = i
    uint64_t* children;             // [capacity] records
    uint16_t* moves;                // [capacity] viriformat move words
    uint32_t* parents;              // [capacity]
    uint32_t* first;                // [nPositions] index of the position's first child
    uint32_t* count;                // [nPositions] number of legal moves
    uint8_t* inCheck;               // [nPositions] side to move in check (mate vs stalemate when count == 0)
    uint32_t* cursor;               // running child count, zero on entry; may end above capacity (= overflow)
    uint32_t capacity;
};

struct PickParams {                 // spx_pick_kernel (spx_movegen.hip): one uniformly random legal move per position
    uint32_t nGames;
    const uint32_t* first;          // per position: block of children (spx_movegen_kernel)
    const uint32_t* count;
    const uint8_t* enable;          // optional per position: 0 = leave this one alone
    const uint64_t* children;       // records as u64[4]
    uint64_t* positions;            // [nGames] records as u64[4]: replaced by the chosen child
    uint64_t* rng;                  // [nGames] splitmix64 state
};

// ---- device-resident self-play (spx_game_step_kernel, spx_movegen.hip): the per-game bookkeeping of
//      src/datagen/datagen.cpp:153-300 for every seat of one half, one wavefront per seat ----
struct SeatState {          // one per seat
    uint32_t plies;         // moves recorded in the current game
    uint32_t win, loss, draw;  // adjudication counters (datagen.cpp:197-199)
    uint32_t startPly;      // Position::plyFromStartpos of the game's initial position
    uint32_t active;        // a game is in progress on this seat
    uint32_t pendingFifty;  // the last move took the halfmove clock to 100: draw unless this ply finds a checkmate ...
    uint32_t reserved;      // ... in which case the result adjudicated with that move (+ 1; 0 = none) stands
};

struct SelfplayCounters {   // one per run, device memory, shared by the halves; a copy travels to the host after every ply
    unsigned long long games, positions, outcomes[3], discarded;
    uint32_t started;       // games counted towards the target (begun and not discarded by the verification filter)
    uint32_t poolCursor;    // openings taken from the pool so far
    uint32_t poolSize;      // openings the host has published so far (ring: entry i lives at i % poolCap)
    uint32_t reserved;
};
static_assert(sizeof(SelfplayCounters) % 8 == 0 && sizeof(SelfplayCounters) / 8 < 62, "spx_game_status_kernel copies it in 64-bit words");

struct GameStepParams {
    uint32_t nSeats;                // seats of this half; every pointer below is already offset to its first seat
    uint32_t seatBase, nSeatsTotal; // global index of the half's first seat; G (slots: seat / G + seat, null slot 2 G)
    const uint32_t* first;          // movegen outputs for the seats' current positions
    const uint32_t* count;
    const uint8_t* inCheck;
    const int32_t* evals;           // per child (side to move of the child)
    const uint16_t* moves;
    const uint64_t* children;       // records as u64[4]
    uint64_t* positions;            // [nSeats] current records
    uint32_t* slots;                // [nSeats] accumulator slot of the current position
    uint64_t* rng;                  // [nSeats] splitmix64 state of the game's move choice
    SeatState* state;               // [nSeats]
    uint64_t* initial;              // [nSeats] records the games started from
    uint32_t* gameMoves;            // [nSeats][maxPlies] viriformat move | recorded score << 16
    uint64_t* keys;                 // [nSeats][maxPlies] keys of the positions played through (repetition detection)
    uint32_t maxPlies;
    int32_t temperature;
    uint32_t targetGames;
    const uint64_t* poolRecords;    // [poolCap] opening records as u64[4]
    const uint64_t* poolSeeds;      // [poolCap]
    uint32_t poolCap;
    SelfplayCounters* counters;
    // The output ring and its write position belong to THIS HALF alone: the halves run concurrently on two streams,
    // and a position shared between them let one half's status kernel publish words the other half had reserved but not yet
    // written. A half's status kernel runs behind its own step kernel, so every word below its snapshot of `streamWords` is there.
    uint32_t* ring;                 // [ringWords] viriformat output (page-locked host memory mapped into the device)
    uint32_t ringWords;
    unsigned long long* streamWords;  // 4-byte words this half has written so far (ring position = mod ringWords)
    uint32_t* updParents;           // [nSeats] the half's materialising update, one record per seat: parent slot (the null
    uint32_t* updChildren;          //          slot for a new game or an idle seat), child slot (the seat's other slot) ...
    uint64_t* updPositions;         // [nSeats] ... and the seat's new current record (empty for an idle seat)
};

// ---- live fixed-node search in the device self-play driver (spx_search_step_kernel, spx_movegen.hip) ----
// What replaces datagen's Searcher::runDatagenSearch (search.cpp:212-239, soft node limit datagen.cpp:78-80) here: iterative
// deepening alpha-beta over the seat's own explicit stack, ONE node expanded per seat and round - the node's children are
// generated by spx_movegen_kernel and evaluated by the fused eval-only update together with every other seat's, exactly
// like the depth-1 driver's plies. The rules (also restated in tests/_search_rules.py, which the games are replayed through):
//   value(child)  = network output of the child from the mover's side, clamped like eval::adjustStatic (eval.cpp:24-27)
//   search(node, depth, alpha, beta, ply): no legal move -> -(kScoreMate - ply) in check, else 0; depth 1 -> the best child
//     value; else the children in the order (value descending, viriformat move word ascending) - at the root of iteration
//     >= 2 the previous iteration's best move first -, fail-soft negamax with cut-off at alpha >= beta
//   iterations 1, 2, ... until the expansions of this search reach the node budget K, the depth reaches kSearchLevels, or
//     the score is decisive (core.h:722-724); the root's own expansion is shared by all iterations
//   K <= 1: the depth-1 policy of spx_game_step_kernel (incl. its temperature), move for move
// No transposition table, no repetition / 50-move detection inside the tree (the game loop's own rules apply to the moves played).
constexpr uint32_t kSearchLevels = 8;       // frames per seat: the root (level 0) .. level 7
constexpr uint32_t kSearchChildren = 224;   // children kept per frame (the legal maximum is 218)
constexpr int32_t kSearchInf = 32767, kSearchMate = 32766;  // core.h:705-706
struct SearchSeat {          // one per seat
    uint32_t top;            // level of the node expanded this round (its children are this round's batch)
    uint32_t iter;           // depth of the running iteration
    uint32_t nodes;          // expansions of this search so far
    int32_t prevBest;        // root child the last completed iteration chose (-1: none yet)
};
struct SearchFrame {         // one per seat and level, 64 bytes
    uint32_t count, depth;   // children; remaining depth (>= 1)
    int32_t alpha, beta, best, bestIdx;
    int32_t cur;             // the child being searched below this frame
    uint32_t reserved;
    uint64_t visited[4];     // children already searched in this visit
};
struct SearchStepParams {
    GameStepParams game;            // (positions / slots: the games' CURRENT positions; first / count / inCheck / evals /
                                    //  moves / children: this round's batch, i.e. the children of every seat's `pending` node)
    uint32_t nodeBudget;            // K
    SearchSeat* seats;              // [nSeats]
    SearchFrame* frames;            // [nSeats][kSearchLevels]
    uint64_t* frameRecords;         // [nSeats][kSearchLevels][kSearchChildren] records as u64[4]
    int32_t* frameValues;           // [nSeats][kSearchLevels][kSearchChildren]
    uint16_t* frameWords;           // [nSeats][kSearchLevels][kSearchChildren] viriformat move words
    uint64_t* pending;              // [nSeats] records: the node each seat expands next (the next move generation's input)
    uint32_t* pendingSlots;         // [nSeats] its accumulator slot
    uint32_t levelSlotBase;         // accumulator slot of (seat, level L >= 1) = levelSlotBase + (L - 1) * nSeatsTotal + seat
    unsigned long long* expansions; // run-wide count of expanded nodes
};

struct ViriExpandParams {          // spx_viri_expand_kernel (spx_movegen.hip)
    const uint8_t* data;            // the viriformat stream
    uint32_t nGames;
    const uint64_t* gameOffset;     // [nGames] byte offset of each game's 32-byte start record
    const uint64_t* outOffset;      // [nGames + 1] index of each game's first output record (prefix sum of its moves)
    uint64_t* out;                  // records as u64[4]
    uint8_t* unfiltered;            // optional, per record: 1 = Marlinformat::push would store it (not in check, the
                                    // played move not noisy: datagen.cpp:254, marlinformat.cpp:31-36), 0 = filtered
    uint32_t* badGames;             // counter: games with a move whose from-square holds no piece of the side to move
};

struct SortParams {
    const uint64_t* positions;  // records as u64[4]
    uint32_t nPositions;
    const uint32_t* nPositionsPtr;  // optional device-resident count (<= nPositions), as in UpdateParams
    bool outOnly;               // large sorts: only the output-bucket order (arena paths: nobody reads the king-bucket order)
    uint8_t* kingKeys;          // [2 * nPositions] scratch
    uint8_t* outKeys;           // [nPositions] scratch
    uint32_t* hist;             // [kHistWords] counts + cursors (layout in spx_kernels.hip); all-zero on entry (large sorts)
    uint32_t* histNext;         // [kHistWords] cleared by this sort for the next large sort
    uint32_t* perspOrder;       // out: [2 * nPositions] perspective ids grouped by king bucket
    uint32_t* posOrder;         // out: [nPositions] position ids grouped by output bucket
};

enum MlpTiling { kMlpTileSorted = 0, kMlpTileShared = 1, kMlpTilePerPosition = 2 };

struct MlpParams {
    uint32_t nPositions;
    const uint32_t* posOrder;   // positions grouped by output bucket
    const uint32_t* hist;       // counts per output bucket at hist[16..23]
    const uint64_t* records;    // kMlpTilePerPosition only: the positions' records (output bucket from the occupancy)
    const uint8_t* ftOut;       // [nPositions][1024]
    const int8_t* l1W;          // device layout [8 buckets][16 ksteps][2 ntiles][64 lanes][16 B]
    const int32_t* l1B;         // [8][32]
    const int32_t* l2W;         // device layout [8 buckets][16 input quartets][64 outputs][4] (relayoutL2)
    const int32_t* l2B;         // [8][64]
    const int32_t* l3W;         // [8][64]
    const int32_t* l3B;         // [8]
    int32_t* out;               // [nPositions]
};

hipError_t launchFt(const FtParams& p, uint32_t gridBlocks, hipStream_t stream);
// one WORKGROUP per perspective (spx_ft_team_kernel): launches with fewer perspectives than wave slots
hipError_t launchFtTeam(const FtParams& p, uint32_t gridBlocks, hipStream_t stream);
hipError_t launchMlp(const MlpParams& p, bool smallL2Weights, MlpTiling tiling, hipStream_t stream);
hipError_t launchSort(const SortParams& p, hipStream_t stream);
hipError_t launchUpdate(const UpdateParams& p, uint32_t gridBlocks, bool splitPerspectives, bool streamAccumulators,
                        hipStream_t stream);
hipError_t launchUpdateObserved(const UpdateParams& p, uint32_t gridBlocks, hipStream_t stream);
hipError_t launchUpdateChain(const ChainParams& p, hipStream_t stream);
hipError_t launchMovegen(const MovegenParams& p, uint32_t gridBlocks, hipStream_t stream);
hipError_t launchViriExpand(const ViriExpandParams& p, hipStream_t stream);
hipError_t launchPick(const PickParams& p, hipStream_t stream);
hipError_t launchGameStep(const GameStepParams& p, hipStream_t stream);
hipError_t launchSearchStep(const SearchStepParams& p, hipStream_t stream);
// hostStatus: device view of page-locked host memory laid out as { SelfplayCounters, u64 streamWords of the half, u64 total }
hipError_t launchGameStatus(const SelfplayCounters* counters, uint32_t* total, const unsigned long long* streamWords,
                            void* hostStatus, hipStream_t stream);
hipError_t launchAdjust(const AdjustParams& p, hipStream_t stream);
hipError_t launchSlotAct(const SlotActParams& p, uint32_t gridBlocks, hipStream_t stream);
uint32_t ftWavesPerBlock();

}  // namespace spx
