// spx_nnue.hpp - C++ mirror of Stormphrax's eval interface on top of the C ABI (spx_nnue.h). Header only.
//
// The reference's seam for this path is the C++ API of src/eval (there is no FFI): free functions in
// src/eval/nnue.h:38-63 and class eval::NnueState (src/eval/nnue_state.h:85-116: reset / push / pop / evaluate /
// evaluateOnce), used by one search thread each over a shared read-only network (src/thread.h:147). The classes below
// keep those names and that discipline - an accumulator *stack* with lazy updates - and map it onto the library's
// arena: stack depth d lives in an arena slot, push() only records the child position, evaluate() materialises ALL
// pending plies in one launch (spx_acc_update_chain_eval: like ensureUpToDate walks forward from the last clean
// ancestor, src/eval/nnue_state.cpp:636-697) and evaluates the top; applyImmediately() advances the root in place from an
// observer delta (datagen's use, src/datagen/datagen.cpp:257-262). Positions cross the boundary as marlinformat PackedBoard
// records (src/datagen/marlinformat.h:32-84 == spx_packed_pos), which Stormphrax already produces.
//
// Errors: the reference's calls cannot fail (asserts only); here a failing library call throws spx_nnue::Error with
// the library's message. Throughput comes from the batched calls (evaluateBatch / the C ABI), not from this stack.
#ifndef SPX_NNUE_HPP
#define SPX_NNUE_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "spx_nnue.h"
#ifdef SPX_NNUE_DEV
#include "spx_nnue_dev.h"  // (Network::synthetic only: the repo's synthetic presets, for tools and tests - libspx_nnue_dev.so)
#endif

namespace spx_nnue {

struct Error : std::runtime_error {
    int status;
    Error(int s, const char* msg) : std::runtime_error(msg ? msg : "spx error"), status(s) {}
};

inline void check(int status) {
    if (status != SPX_OK) throw Error(status, spx_last_error());
}

// eval::init / getNetwork / defaultNetworkName (src/eval/nnue.h:38-44): an immutable network shared by all states
class Network {
public:
    Network(const void* blob, size_t nbytes) {
        check(spx_net_load(blob, nbytes, &net_));
    }
#ifdef SPX_NNUE_DEV
    static Network synthetic(int preset, uint64_t seed = 20260927) {  // 0 tame, 1 wild, 2 extreme (spx_synth_net)
        std::vector<unsigned char> buf(spx_synth_net_bytes());
        check(spx_synth_net(seed, preset, buf.data(), buf.size()));
        return Network(buf.data(), buf.size());
    }
#endif
    Network(Network&& o) noexcept : net_(o.net_) {
        o.net_ = nullptr;
    }
    Network(const Network&) = delete;
    Network& operator=(const Network&) = delete;
    ~Network() {
        spx_net_free(net_);
    }
    const char* name() const {
        return spx_net_name(net_);
    }
    const spx_net* handle() const {
        return net_;
    }

private:
    spx_net* net_ = nullptr;
};

// eval::NnueState (src/eval/nnue_state.h:85-116)
class NnueState {
public:
    static constexpr uint32_t kMaxDepth = 256;  // the reference's accumulator stack holds 256 entries (nnue_state.h:104)

    // `options`: "name=value,..." tuning knobs of this context (spx_ctx_create_opts; nullptr = the defaults)
    explicit NnueState(const Network& net, int device = 0, size_t maxBatch = 4096, const char* options = nullptr) {
        check(spx_ctx_create_opts(net.handle(), device, maxBatch, 0u, options, &ctx_));
        try {
            check(spx_acc_reserve(ctx_, kMaxDepth + 1));  // one slot per stack entry + a spare for applyImmediately
        } catch (...) {
            spx_ctx_destroy(ctx_);
            throw;
        }
        stack_.reserve(kMaxDepth);
        slotOf_.resize(kMaxDepth);
        for (uint32_t d = 0; d < kMaxDepth; ++d) slotOf_[d] = d;
        spare_ = kMaxDepth;
    }
    NnueState(const NnueState&) = delete;
    NnueState& operator=(const NnueState&) = delete;
    ~NnueState() {
        spx_ctx_destroy(ctx_);
    }

    // NnueState::reset (nnue_state.cpp:539-560): full refresh of both perspectives, stack back to depth 0
    void reset(const spx_packed_pos& pos) {
        stack_.assign(1, pos);
        clean_ = 0;
        check(spx_acc_refresh(ctx_, &pos, &slotOf_[0], 1));
    }
    // NnueState::push + Position::applyMove (src/thread.cpp:46-67): `child` is the position after the move. Nothing is
    // computed yet - the entry is dirty until the next evaluate(), exactly like the reference's lazy UpdateContext
    void push(const spx_packed_pos& child) {
        if (stack_.empty()) throw Error(SPX_ERR_INVALID_ARG, "NnueState::push before reset");
        if (stack_.size() >= kMaxDepth) throw Error(SPX_ERR_CAPACITY, "NnueState: accumulator stack overflow");
        stack_.push_back(child);
    }
    // NnueState::pop (src/thread.h:116-122): the parent's accumulators are still materialised in their slot
    void pop() {
        if (stack_.size() <= 1) throw Error(SPX_ERR_INVALID_ARG, "NnueState::pop at the root");
        stack_.pop_back();
        if (clean_ > stack_.size() - 1) clean_ = uint32_t(stack_.size() - 1);
    }
    // NnueState::evaluate (nnue_state.cpp:598-610): ensureUpToDate, then the network on the top of the stack. The side
    // to move comes from the position record.
    int32_t evaluate() {
        if (stack_.empty()) throw Error(SPX_ERR_INVALID_ARG, "NnueState::evaluate before reset");
        const uint32_t top = uint32_t(stack_.size() - 1);
        int32_t out = 0;
        if (clean_ < top) {  // pending plies clean_ + 1 .. top: one launch for the whole path, fused with the evaluation
            check(spx_acc_update_chain_eval(ctx_, slotOf_[clean_], &slotOf_[clean_ + 1], &stack_[clean_ + 1], top - clean_, &out));
            clean_ = top;
            return out;
        }
        check(spx_acc_eval(ctx_, &slotOf_[top], 1, &out));
        return out;
    }
    // NnueState::applyImmediately (nnue_state.cpp:572-591; datagen.cpp:257-262): the move has been made with an observer -
    // `delta` is the UpdateContext it captured (spx_pos_apply_uci_observed, or the host's own BoardObserver), `child` the
    // position after it. The TOP of the stack becomes `child`, its accumulators updated at once from the old top's (no push:
    // the stack depth is unchanged, nothing stays pending).
    void applyImmediately(const spx_move_delta& delta, const spx_packed_pos& child) {
        if (stack_.empty()) throw Error(SPX_ERR_INVALID_ARG, "NnueState::applyImmediately before reset");
        evaluateIfPending();
        const uint32_t top = uint32_t(stack_.size() - 1);
        check(spx_acc_update_observed(ctx_, &slotOf_[top], &spare_, &child, &delta, 1, nullptr));
        std::swap(slotOf_[top], spare_);  // the old top's slot is the next spare
        stack_[top] = child;
    }
    // NnueState::evaluateOnce (nnue_state.cpp:612-634): from scratch, no state touched
    int32_t evaluateOnce(const spx_packed_pos& pos) {
        int32_t out = 0;
        check(spx_eval_full(ctx_, &pos, 1, &out));
        return out;
    }
    // the batched form of evaluateOnce: what the GPU is for
    std::vector<int32_t> evaluateBatch(const spx_packed_pos* positions, size_t n) {
        std::vector<int32_t> out(n);
        check(spx_eval_full(ctx_, positions, n, out.data()));
        return out;
    }
    // eval::staticEvalOnce (src/eval/eval.cpp:109-112): + contempt[stm], clamp to +-24999
    int32_t staticEvalOnce(const spx_packed_pos& pos, int32_t contemptBlack = 0, int32_t contemptWhite = 0) {
        int32_t v = evaluateOnce(pos);
        spx_adjust_params params;
        spx_adjust_defaults(&params);
        params.contempt[0] = contemptBlack;
        params.contempt[1] = contemptWhite;
        params.stages = SPX_ADJUST_STATIC;
        check(spx_adjust(ctx_, &pos, 1, &params, nullptr, &v));
        return v;
    }
    // A whole recorded search tree at once (BASELINE config 3): node k > 0 was reached from node parents[k] < k, node 0 is
    // the root; returns NnueState::evaluate at evalNodes. Levels are batched on the device (spx_acc_replay_tree); the
    // stack of this object is left at the root.
    std::vector<int32_t> replayTree(const std::vector<spx_packed_pos>& positions, const std::vector<uint32_t>& parents,
                                    const std::vector<uint32_t>& evalNodes, double* gpuMs = nullptr) {
        if (positions.empty() || positions.size() != parents.size()) throw Error(SPX_ERR_INVALID_ARG, "replayTree: sizes");
        std::vector<int32_t> out(evalNodes.size());
        check(spx_acc_replay_tree(ctx_, positions.data(), parents.data(), positions.size(), evalNodes.data(), evalNodes.size(),
                                  out.data(), gpuMs));
        stack_.assign(1, positions[0]);
        clean_ = 0;
        return out;
    }
    size_t depth() const {
        return stack_.empty() ? 0 : stack_.size() - 1;
    }
    size_t pending() const {  // stack entries whose accumulators are not materialised yet
        return stack_.empty() ? 0 : stack_.size() - 1 - clean_;
    }
    const spx_packed_pos& position() const {
        return stack_.back();
    }
    spx_ctx* context() {
        return ctx_;
    }

private:
    void evaluateIfPending() {
        const uint32_t top = uint32_t(stack_.size() - 1);
        if (clean_ < top) {
            check(spx_acc_update_chain_eval(ctx_, slotOf_[clean_], &slotOf_[clean_ + 1], &stack_[clean_ + 1], top - clean_, nullptr));
            clean_ = top;
        }
    }
    spx_ctx* ctx_ = nullptr;
    std::vector<spx_packed_pos> stack_;  // position at every stack depth
    std::vector<uint32_t> slotOf_;       // arena slot of every stack depth (a permutation: applyImmediately swaps with the spare)
    uint32_t spare_ = 0;
    uint32_t clean_ = 0;                 // deepest stack entry whose slot holds up-to-date accumulators
};

// One context per GPU inside this process (spx_group): evaluateBatch cuts a batch into contiguous shards, one
// per device, each evaluated on its own host thread - same results as one NnueState evaluating the whole batch.
class DeviceGroup {
public:
    // devices: HIP ordinals; empty = every visible device
    DeviceGroup(const Network& net, const std::vector<int>& devices, size_t maxBatchPerDevice) {
        check(spx_group_create(net.handle(), devices.empty() ? nullptr : devices.data(), devices.size(), maxBatchPerDevice, 0,
                               &group_));
    }
    DeviceGroup(const DeviceGroup&) = delete;
    DeviceGroup& operator=(const DeviceGroup&) = delete;
    ~DeviceGroup() {
        spx_group_destroy(group_);
    }
    size_t size() const {
        return spx_group_size(group_);
    }
    std::vector<int32_t> evaluateBatch(const spx_packed_pos* positions, size_t n) {
        std::vector<int32_t> out(n);
        check(spx_group_eval_full(group_, positions, n, out.data()));
        return out;
    }
    spx_group* handle() {
        return group_;
    }

private:
    spx_group* group_ = nullptr;
};

}  // namespace spx_nnue

#endif  // SPX_NNUE_HPP
