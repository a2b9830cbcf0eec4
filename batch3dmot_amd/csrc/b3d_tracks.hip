// Hierarchical track clustering of the greedy edges (reference predict.py:262-375, mode "hier"): HOST code.
// The procedure is a sequential greedy merge -- every decision depends on the clusters the higher-scoring edges
// have already formed -- over at most two edges per detection of one scene, so it belongs on the host; the
// reference walks Python dictionaries and lists (list.insert(0, ..), list concatenation, max(keys)) and prints
// several lines per edge, here clusters are doubly linked lists with O(1) head / tail tests and splices.
#include <algorithm>
#include <numeric>
#include <vector>
#include "b3d_common.hpp"

extern "C" int b3d_tracks_from_edges(const int64_t* pairs, const double* scores, int64_t M, const int64_t* node_class,
                                     int64_t N, const double* join_threshold, int32_t num_classes, int64_t* track_nodes,
                                     int64_t* track_ptr, int64_t* n_tracks) {
  using namespace b3d;
  B3D_REQUIRE(M == 0 || (pairs && scores), "b3d_tracks_from_edges: null edge list");
  B3D_REQUIRE(node_class && join_threshold && track_nodes && track_ptr && n_tracks, "b3d_tracks_from_edges: null argument");
  B3D_REQUIRE(M >= 0 && N > 0 && num_classes > 0, "b3d_tracks_from_edges: M %lld, N %lld", (long long)M, (long long)N);
  for (int64_t e = 0; e < M; ++e)
    B3D_REQUIRE(pairs[2 * e] != pairs[2 * e + 1], "b3d_tracks_from_edges: edge %lld is a self loop (edges link a detection to a LATER one)", (long long)e);
  for (int64_t e = 0; e < M; ++e)
    B3D_REQUIRE(pairs[2 * e] >= 0 && pairs[2 * e] < N && pairs[2 * e + 1] >= 0 && pairs[2 * e + 1] < N,
                "b3d_tracks_from_edges: edge %lld has a node id outside [0, %lld)", (long long)e, (long long)N);
  for (int64_t n = 0; n < N; ++n)
    B3D_REQUIRE(node_class[n] >= 0 && node_class[n] < num_classes, "b3d_tracks_from_edges: class of node %lld outside [0, %d)",
                (long long)n, (int)num_classes);
  // pred_edges_dict: a repeated (j, i) keeps its first position and takes the last score (dict semantics, :290)
  std::vector<int64_t> first(M);
  std::vector<double> val(scores, scores + M);
  {
    std::vector<int64_t> idx(M);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
      return pairs[2 * a] != pairs[2 * b] ? pairs[2 * a] < pairs[2 * b] : pairs[2 * a + 1] < pairs[2 * b + 1];
    });
    for (int64_t k = 0; k < M;) {
      int64_t k2 = k;
      while (k2 < M && pairs[2 * idx[k2]] == pairs[2 * idx[k]] && pairs[2 * idx[k2] + 1] == pairs[2 * idx[k] + 1]) ++k2;
      for (int64_t t = k; t < k2; ++t) first[idx[t]] = idx[k];
      val[idx[k]] = scores[idx[k2 - 1]];
      k = k2;
    }
  }
  std::vector<int64_t> order;
  for (int64_t e = 0; e < M; ++e)
    if (first[e] == e) order.push_back(e);
  // sorted(items, key=score, reverse=True): stable, ties keep insertion order (:291)
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return val[a] > val[b]; });

  std::vector<int64_t> cluster(N, -1), next(N, -1), prev(N, -1);      // vis / list links
  struct Cl { int64_t head, tail, born; bool alive; };
  std::vector<Cl> cl;
  int64_t born = 0;
  for (int64_t e : order) {
    const int64_t j = pairs[2 * e], i = pairs[2 * e + 1];
    const double score = val[e];
    const int64_t cj = cluster[j], ci = cluster[i];
    if (cj < 0 && ci < 0) {                                   // unconstrained edge: a new cluster [j, i]
      cl.push_back(Cl{j, i, born++, true});
      cluster[j] = cluster[i] = (int64_t)cl.size() - 1;
      next[j] = i; prev[i] = j;
    } else if (cj < 0) {                                      // preceding edge: only in front of the cluster's first node
      if (cl[ci].head != i) continue;
      next[j] = i; prev[i] = j; cl[ci].head = j; cluster[j] = ci;
    } else if (ci < 0) {                                      // succeeding edge: only behind the cluster's last node
      if (cl[cj].tail != j) continue;
      next[j] = i; prev[i] = j; cl[cj].tail = i; cluster[i] = cj;
    } else {                                                  // both visited: join tail of cluster(j) to head of cluster(i)
      if (cj == ci) continue;                                 // (the reference would duplicate the list; edges go forward in time, so it never happens)
      if (!(cl[cj].tail == j && cl[ci].head == i && score > join_threshold[node_class[i]])) continue;
      next[j] = i; prev[i] = j;
      for (int64_t n = cl[ci].head; n >= 0; n = next[n]) cluster[n] = cj;
      cl[cj].tail = cl[ci].tail;
      cl[ci].alive = false;
    }
  }
  // [v for k, v in clusters.items()]: surviving clusters in creation order
  int64_t nt = 0, pos = 0;
  track_ptr[0] = 0;
  for (const Cl& c : cl) {
    if (!c.alive) continue;
    for (int64_t n = c.head; n >= 0; n = next[n]) track_nodes[pos++] = n;
    track_ptr[++nt] = pos;
  }
  *n_tracks = nt;
  return B3D_OK;
}
