// Internal (non-ABI) declarations shared between the host translation units of libspx_nnue.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

#include "spx_arch.h"

namespace spx {

// spx_synth.cpp
size_t synthNetBytes();
bool synthNet(uint64_t seed, int preset, void* buf, size_t n);
uint64_t fnv1a64(const void* data, size_t n);

// spx_api.cpp: capacity of a context (spx_ctx is opaque outside spx_api.cpp)
struct spx_ctx_fwd;
}  // namespace spx
struct spx_ctx;
namespace spx {
size_t ctxMaxBatch(const spx_ctx* ctx);
int ctxDevice(const spx_ctx* ctx);
void* ctxStream(const spx_ctx* ctx);  // the context's own hipStream_t (what a NULL stream argument means)
int64_t ctxSelfplayOption(const spx_ctx* ctx, int which);  // 0 selfplay_graph, 1 selfplay_graph_plies, 2 selfplay_trace (spx_ctx_set_option)
uint8_t* ctxSlotRecords(const spx_ctx* ctx);  // device pointer: the arena's [nSlots][32] record store (after spx_acc_reserve)
// lanes: see spx_api.cpp (two scratch sets + streams; big kernels chained by events)
// gates = false: the lane's big kernels are NOT chained to the other lane's by events (a stream that is being captured
// into a graph cannot wait on events recorded outside the capture)
int ctxLaneBegin(spx_ctx* ctx, int laneIndex, void** stream, bool gates = true);
void ctxLaneEnd(spx_ctx* ctx, int laneIndex);

// error plumbing (spx_api): thread-local last error string, returned by spx_last_error()
void setError(const std::string& msg);

}  // namespace spx
