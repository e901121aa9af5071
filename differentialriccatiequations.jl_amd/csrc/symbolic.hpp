// Host-side symbolic analysis for the multifrontal sparse LU of the shifted pencil
//   M(mu) = cA * A' + cE * E'      (pattern fixed for the whole run, values change per shift).
// Nested-dissection ordering (BFS level-set separators), separator tree, frontal index sets,
// assembly and extend-add maps, level schedule.  Pure C++ (no HIP) so it is unit-testable on CPU.
#pragma once
#include <cstdint>
#include <vector>

namespace dre {

struct Symbolic {
    int n = 0;
    // permutation: perm[new] = old, iperm[old] = new
    std::vector<int> perm, iperm;
    // permuted union pattern in CSR (rows sorted), including the diagonal
    std::vector<int> ptr, idx;
    // separator tree, nodes in postorder (children before parents)
    int nnodes = 0;
    std::vector<int> first, size;          // node t owns permuted indices [first, first+size)
    std::vector<int> parent;               // -1 for roots
    std::vector<int> level;                // depth from the root (roots = 0)
    std::vector<int> child_ptr, child_idx; // children lists
    std::vector<int> bptr;                 // boundary index sets: B_t = bidx[bptr[t] .. bptr[t+1])
    std::vector<int> bidx;                 // sorted permuted indices > last index of t
    std::vector<int> cmap_ptr, cmap;       // per node t (as a CHILD): local index in the parent's front of each B_t entry
    std::vector<int64_t> front_off;        // offset of front t in the fronts slab (f_t*f_t entries, column-major)
    std::vector<int64_t> inv_off;          // offset of the s_t*s_t inverse-diagonal-block storage
    std::vector<int64_t> upd_off;          // offset (in rows) of node t's update rows in the solve workspace
    int64_t fronts_size = 0, inv_size = 0, upd_rows = 0;
    std::vector<int64_t> asm_dest;         // per nonzero of the permuted pattern: destination in the fronts slab
    int nlevels = 0;
    std::vector<int> lvl_ptr, lvl_nodes;   // nodes grouped by level (level 0 = roots)
    int max_front = 0, max_sep = 0;
    int64_t factor_nnz = 0;                // sum over nodes of s*(s + 2b): entries of L and U touched by a solve

    int fsize(int t) const { return size[t] + (bptr[t + 1] - bptr[t]); }
};

// pattern: CSR of an n x n matrix (any pattern; it is symmetrised internally).  leaf_size: stop dissecting below.
Symbolic symbolic_analyze(int n, const std::vector<int>& ptr, const std::vector<int>& idx, int leaf_size);

}  // namespace dre
