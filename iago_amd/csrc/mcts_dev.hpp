// mcts_dev.hpp -- device helpers of the tree kernels (mcts_kernels.hip) that the persistent search kernel
// (search_kernel.hip) shares: node initialisation and the argmax butterfly of Node.select (MCTS.py:39-49).
#pragma once
#include "abi_common.hpp"
#include "othello_dev.hpp"

#include <math.h>

namespace iago_mcts {
using namespace iago;

constexpr int BLOCK = 256;
constexpr int MAX_DEPTH = 512; // bound on the descent (a path adds a node per expansion)

typedef iago_mcts_tree Tree;

__device__ __forceinline__ void init_node(const Tree &T, int64_t i, int parent, int action, float p)
{
    T.nodes[i].parent = parent;
    T.nodes[i].first_child = -1;
    T.nodes[i].n_children = 0;
    T.nodes[i].action = (int8_t)action;
    T.nodes[i].n_visits = 0;
    T.nodes[i].q = 0.0f;
    T.nodes[i].p = p;
    if (T.has_v)
        T.nodes[i].v = __builtin_nanf(""); // value_func(node) not evaluated yet
}

template <int CTRL>
__device__ __forceinline__ void argmax_step(double &v, int &idx)
{
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)bits), hi = dpp_u32<CTRL>((uint32_t)(bits >> 32));
    const double ov = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    const int oi = (int)dpp_u32<CTRL>((uint32_t)idx);
    const bool take = (ov > v) || (ov == v && oi < idx);
    v = take ? ov : v;
    idx = take ? oi : idx;
}

// the same step carrying four payload words of the winner
template <int CTRL>
__device__ __forceinline__ void argmax_step_payload(double &v, int &idx, uint32_t (&pl)[4])
{
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = dpp_u32<CTRL>((uint32_t)bits), hi = dpp_u32<CTRL>((uint32_t)(bits >> 32));
    const double ov = __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    const int oi = (int)dpp_u32<CTRL>((uint32_t)idx);
    const bool take = (ov > v) || (ov == v && oi < idx);
    v = take ? ov : v;
    idx = take ? oi : idx;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t o = dpp_u32<CTRL>(pl[i]);
        pl[i] = take ? o : pl[i];
    }
}

} // namespace iago_mcts
