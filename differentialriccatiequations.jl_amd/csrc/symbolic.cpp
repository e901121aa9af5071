// Nested-dissection ordering + symbolic multifrontal analysis (host, pure C++).
#include "symbolic.hpp"

#include <algorithm>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace dre {
namespace {

struct Dissector {
    int n;
    const std::vector<int>& aptr;
    const std::vector<int>& aidx;   // symmetric adjacency without self loops
    int leaf_size;
    std::vector<int> mark;          // subset stamp per vertex
    std::vector<int> lev;           // BFS level scratch
    int stamp = 0;
    // output
    std::vector<std::vector<int>> node_verts;
    std::vector<int> node_parent;
    std::vector<std::vector<int>> node_children;

    Dissector(int n_, const std::vector<int>& p, const std::vector<int>& i, int leaf)
        : n(n_), aptr(p), aidx(i), leaf_size(leaf), mark(n_, -1), lev(n_, -1) {}

    int new_node(std::vector<int>&& verts, const std::vector<int>& children) {
        int id = (int)node_verts.size();
        node_verts.push_back(std::move(verts));
        node_parent.push_back(-1);
        node_children.push_back(children);
        for (int c : children) node_parent[c] = id;
        return id;
    }

    // BFS inside the subset stamped `st`, starting at s; fills order (visited vertices) and lev[]; returns #levels
    int bfs(int s, int st, std::vector<int>& order) {
        order.clear();
        order.push_back(s);
        lev[s] = 0;
        mark[s] = st + 1;   // visited stamp = st+1
        size_t head = 0;
        int maxlev = 0;
        while (head < order.size()) {
            int u = order[head++];
            for (int p = aptr[u]; p < aptr[u + 1]; ++p) {
                int v = aidx[p];
                if (mark[v] == st) {
                    mark[v] = st + 1;
                    lev[v] = lev[u] + 1;
                    maxlev = std::max(maxlev, lev[v]);
                    order.push_back(v);
                }
            }
        }
        return maxlev + 1;
    }
    void restamp(const std::vector<int>& verts, int st) { for (int v : verts) mark[v] = st; }

    // depth cap: below the depth of a balanced tree (log2(n / leaf_size)) a subdomain of up to four leaf sizes becomes a (dense) leaf, so
    // that the few deep branches of an unbalanced dissection do not add tree levels — every level is one launch per sweep direction
    // (sparse.hip), whatever the number of its nodes
    int depth_cap = 1 << 30;
    int dissect(std::vector<int>& verts, int depth = 0) {
        if ((int)verts.size() <= leaf_size || (depth >= depth_cap && (int)verts.size() <= 4 * leaf_size)) return new_node(std::move(verts), {});
        // stamps: st = member, st+1 = visited
        int st = (stamp += 2);
        restamp(verts, st);
        std::vector<int> order;
        // connected components
        std::vector<std::vector<int>> comps;
        for (int v : verts) {
            if (mark[v] == st) {
                bfs(v, st, order);
                comps.push_back(order);
            }
        }
        if (comps.size() > 1) {
            std::sort(comps.begin(), comps.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
            std::vector<int> g1, g2;
            for (auto& c : comps) {
                auto& g = (g1.size() <= g2.size()) ? g1 : g2;
                g.insert(g.end(), c.begin(), c.end());
            }
            int c1 = dissect(g1, depth + 1);
            int c2 = dissect(g2, depth + 1);
            return new_node({}, {c1, c2});   // empty separator
        }
        // pseudo-peripheral vertex: two sweeps
        st = (stamp += 2);
        restamp(verts, st);
        bfs(verts[0], st, order);
        int far = order.back();
        st = (stamp += 2);
        restamp(verts, st);
        int nlev = bfs(far, st, order);
        if (nlev < 3) return new_node(std::move(verts), {});   // no usable level structure: dense leaf
        std::vector<int> cnt(nlev, 0);
        for (int v : order) cnt[lev[v]]++;
        std::vector<int> cum(nlev + 1, 0);
        for (int l = 0; l < nlev; ++l) cum[l + 1] = cum[l] + cnt[l];
        int best = 1;
        long bestscore = -1;
        const int tot = (int)order.size();
        for (int l = 1; l <= nlev - 2; ++l) {
            int below = cum[l], above = tot - cum[l + 1];
            long score = (long)std::max(below, above) + cnt[l];
            if (bestscore < 0 || score < bestscore) { bestscore = score; best = l; }
        }
        std::vector<int> sep, p1, p2;
        for (int v : order) {
            if (lev[v] < best) p1.push_back(v);
            else if (lev[v] > best) p2.push_back(v);
            else {
                bool touches_above = false;
                for (int p = aptr[v]; p < aptr[v + 1]; ++p) {
                    int u = aidx[p];
                    if (mark[u] == st + 1 && lev[u] == best + 1) { touches_above = true; break; }
                }
                (touches_above ? sep : p1).push_back(v);
            }
        }
        if (p1.empty() || p2.empty()) return new_node(std::move(verts), {});
        verts.clear();
        verts.shrink_to_fit();
        int c1 = dissect(p1, depth + 1);
        int c2 = dissect(p2, depth + 1);
        return new_node(std::move(sep), {c1, c2});
    }
};

}  // namespace

Symbolic symbolic_analyze(int n, const std::vector<int>& ptr, const std::vector<int>& idx, int leaf_size) {
    Symbolic S;
    S.n = n;
    if (n == 0) return S;
    // symmetric adjacency without self loops
    std::vector<std::vector<int>> adj(n);
    for (int i = 0; i < n; ++i)
        for (int p = ptr[i]; p < ptr[i + 1]; ++p) {
            int j = idx[p];
            if (j < 0 || j >= n) throw std::runtime_error("symbolic_analyze: column index out of range");
            if (j != i) { adj[i].push_back(j); adj[j].push_back(i); }
        }
    std::vector<int> aptr(n + 1, 0), aidx;
    for (int i = 0; i < n; ++i) {
        auto& a = adj[i];
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
        aptr[i + 1] = aptr[i] + (int)a.size();
    }
    aidx.reserve(aptr[n]);
    for (int i = 0; i < n; ++i) aidx.insert(aidx.end(), adj[i].begin(), adj[i].end());

    Dissector D(n, aptr, aidx, std::max(1, leaf_size));
    {
        int dc = 0;
        while (((long)std::max(1, leaf_size) << (dc + 1)) <= (long)n) ++dc;        // floor(log2(n / leaf_size))
        D.depth_cap = std::max(1, dc);
    }
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    D.dissect(all);

    // Supernode amalgamation (2:1): every tree level costs one launch per sweep direction and one per factorisation whatever its width
    // (sparse.hip), and on these meshes the levels are latency bound, so two levels of bisection are merged into one four-way node: a node
    // at odd depth hands its separator (a leaf: its whole subdomain) to its parent, its children become children of the parent.  The merged
    // front treats the (structurally zero) coupling between the two absorbed separators as dense — a few percent more factor entries for half
    // the launches.  Vertices of the absorbed nodes come first inside the merged node (they are eliminated first).
    // MEASURED (round 2) and therefore OFF by default: n = 20209 goes from 12 to 5 levels and from 132 to 236 ms (factor entries +58 %, fronts
    // of up to 187 pivots: their factorisation and their sweeps cost far more than the launches saved); n = 5177: 137.7 -> 142.5 ms.
    {
        int amalg = 0;
        // (2:1 supernode amalgamation of the separator tree: built and measured in round 2 — 12 -> 5 levels at n = 20209 but +58 % factor entries,
        //  132 -> 236 ms; kept as code for deep narrow trees, switched off)
        if (amalg) {
            const int T0 = (int)D.node_verts.size();
            int root = -1;
            for (int t = 0; t < T0; ++t) if (D.node_parent[t] < 0) root = t;       // the dissection returns one root (created last)
            std::vector<std::vector<int>> nverts; std::vector<int> nparent; std::vector<std::vector<int>> nchildren;
            // recursive build in postorder (explicit stack: depth is only ~log2 n, recursion is fine)
            struct Rec {
                Dissector& D; std::vector<std::vector<int>>& nv; std::vector<int>& np; std::vector<std::vector<int>>& nc;
                int build(int t) {
                    std::vector<int> verts, kids;
                    for (int c : D.node_children[t]) {
                        verts.insert(verts.end(), D.node_verts[c].begin(), D.node_verts[c].end());     // absorbed (odd depth)
                        for (int g : D.node_children[c]) kids.push_back(build(g));
                    }
                    verts.insert(verts.end(), D.node_verts[t].begin(), D.node_verts[t].end());
                    const int id = (int)nv.size();
                    nv.push_back(std::move(verts)); np.push_back(-1); nc.push_back(kids);
                    for (int k : kids) np[(size_t)k] = id;
                    return id;
                }
            } rec{D, nverts, nparent, nchildren};
            if (root >= 0) {
                rec.build(root);
                D.node_verts = std::move(nverts); D.node_parent = std::move(nparent); D.node_children = std::move(nchildren);
            }
        }
    }
    // nodes were created in postorder already (children before parents)
    const int T = (int)D.node_verts.size();
    S.nnodes = T;
    S.first.resize(T); S.size.resize(T); S.parent = D.node_parent; S.level.assign(T, 0);
    S.perm.resize(n); S.iperm.resize(n);
    int counter = 0;
    for (int t = 0; t < T; ++t) {
        S.first[t] = counter;
        S.size[t] = (int)D.node_verts[t].size();
        for (int v : D.node_verts[t]) { S.perm[counter] = v; S.iperm[v] = counter; ++counter; }
    }
    if (counter != n) throw std::runtime_error("symbolic_analyze: ordering lost vertices");
    for (int t = T - 1; t >= 0; --t) S.level[t] = S.parent[t] < 0 ? 0 : S.level[S.parent[t]] + 1;
    S.child_ptr.assign(T + 1, 0);
    for (int t = 0; t < T; ++t) S.child_ptr[t + 1] = S.child_ptr[t] + (int)D.node_children[t].size();
    for (int t = 0; t < T; ++t) for (int c : D.node_children[t]) S.child_idx.push_back(c);

    // node of each permuted index
    std::vector<int> node_of(n);
    for (int t = 0; t < T; ++t) for (int i = 0; i < S.size[t]; ++i) node_of[S.first[t] + i] = t;

    // permuted union pattern (with diagonal), rows sorted
    S.ptr.assign(n + 1, 0);
    {
        std::vector<std::vector<int>> rows(n);
        for (int i = 0; i < n; ++i) {
            int pi = S.iperm[i];
            rows[pi].push_back(pi);
            for (int p = ptr[i]; p < ptr[i + 1]; ++p) rows[pi].push_back(S.iperm[idx[p]]);
        }
        for (int i = 0; i < n; ++i) {
            auto& r = rows[i];
            std::sort(r.begin(), r.end());
            r.erase(std::unique(r.begin(), r.end()), r.end());
            S.ptr[i + 1] = S.ptr[i] + (int)r.size();
        }
        S.idx.reserve(S.ptr[n]);
        for (int i = 0; i < n; ++i) S.idx.insert(S.idx.end(), rows[i].begin(), rows[i].end());
    }

    // boundary sets (postorder)
    std::vector<std::vector<int>> B(T);
    for (int t = 0; t < T; ++t) {
        const int last = S.first[t] + S.size[t];   // one past
        std::vector<int>& b = B[t];
        for (int i = S.first[t]; i < last; ++i) {
            int v = S.perm[i];
            for (int p = aptr[v]; p < aptr[v + 1]; ++p) {
                int j = S.iperm[aidx[p]];
                if (j >= last) b.push_back(j);
            }
        }
        for (int c : D.node_children[t])
            for (int j : B[c]) {
                if (j >= last) b.push_back(j);
                else if (j < S.first[t]) throw std::runtime_error("symbolic_analyze: separator property violated");
            }
        std::sort(b.begin(), b.end());
        b.erase(std::unique(b.begin(), b.end()), b.end());
        // every boundary index must belong to an ancestor
        for (int j : b) {
            int a = node_of[j], u = S.parent[t];
            while (u >= 0 && u != a) u = S.parent[u];
            if (u < 0) throw std::runtime_error("symbolic_analyze: boundary index outside the ancestor chain");
        }
    }
    S.bptr.assign(T + 1, 0);
    for (int t = 0; t < T; ++t) S.bptr[t + 1] = S.bptr[t] + (int)B[t].size();
    S.bidx.reserve(S.bptr[T]);
    for (int t = 0; t < T; ++t) S.bidx.insert(S.bidx.end(), B[t].begin(), B[t].end());

    // offsets
    S.front_off.resize(T); S.inv_off.resize(T); S.upd_off.resize(T);
    int64_t fo = 0, io = 0, uo = 0;
    for (int t = 0; t < T; ++t) {
        const int64_t s = S.size[t], b = (int64_t)B[t].size(), f = s + b;
        S.front_off[t] = fo; fo += f * f;
        S.inv_off[t] = io; io += s * s;
        S.upd_off[t] = uo; uo += b;
        S.max_front = std::max<int>(S.max_front, (int)f);
        S.max_sep = std::max<int>(S.max_sep, (int)s);
        S.factor_nnz += s * s + 2 * s * b;
    }
    S.fronts_size = fo; S.inv_size = io; S.upd_rows = uo;

    auto local_index = [&](int t, int x) -> int {
        const int f0 = S.first[t], s = S.size[t];
        if (x >= f0 && x < f0 + s) return x - f0;
        auto it = std::lower_bound(B[t].begin(), B[t].end(), x);
        if (it == B[t].end() || *it != x) throw std::runtime_error("symbolic_analyze: index missing from a front");
        return s + (int)(it - B[t].begin());
    };

    // child -> parent maps
    S.cmap_ptr.assign(T + 1, 0);
    for (int t = 0; t < T; ++t) S.cmap_ptr[t + 1] = S.cmap_ptr[t] + (int)B[t].size();
    S.cmap.assign(S.cmap_ptr[T], -1);
    for (int t = 0; t < T; ++t) {
        int p = S.parent[t];
        if (p < 0) continue;
        for (size_t i = 0; i < B[t].size(); ++i) S.cmap[S.cmap_ptr[t] + i] = local_index(p, B[t][i]);
    }

    // assembly destinations
    S.asm_dest.resize(S.idx.size());
    for (int i = 0; i < n; ++i)
        for (int p = S.ptr[i]; p < S.ptr[i + 1]; ++p) {
            int j = S.idx[p];
            int t = node_of[std::min(i, j)];
            int64_t f = S.size[t] + (int64_t)B[t].size();
            S.asm_dest[p] = S.front_off[t] + local_index(t, i) + (int64_t)local_index(t, j) * f;
        }

    // level schedule
    int nl = 0;
    for (int t = 0; t < T; ++t) nl = std::max(nl, S.level[t] + 1);
    S.nlevels = nl;
    S.lvl_ptr.assign(nl + 1, 0);
    for (int t = 0; t < T; ++t) S.lvl_ptr[S.level[t] + 1]++;
    for (int l = 0; l < nl; ++l) S.lvl_ptr[l + 1] += S.lvl_ptr[l];
    S.lvl_nodes.resize(T);
    {
        std::vector<int> pos(S.lvl_ptr.begin(), S.lvl_ptr.end() - 1);
        for (int t = 0; t < T; ++t) S.lvl_nodes[pos[S.level[t]]++] = t;
    }
    return S;
}

}  // namespace dre
