// Edge phase of a CausalMessagePassing layer at the camera+LiDAR+radar widths, hoisted form (b3d_hoist.hpp), on the
// fragment stream of b3d_estream.hpp: forward (clr_att_gnn.py:302-334 without the node columns of the three first layers,
// which arrive as gathered rows of the per-node table T) and its data gradient.  Same arguments, same results (up to fp32
// summation order inside a Linear: 32-wide k groups instead of whole rows) as mp_edge_fwd_h_kernel / mp_edge_bwd_h_kernel,
// which remain the kernels of the poses-only model (its weights are resident in LDS).
//
// Every per-edge buffer these kernels touch must hold round_up(E, 64) rows: rows past the end are computed on the last
// edge and STORED (into the padding), so that the number of vector-memory instructions between two rendezvous is a
// compile-time constant (b3d_estream.hpp, Ring).
#pragma once
#include "b3d_estream.hpp"
#include "b3d_hoist.hpp"

namespace b3d {
namespace es {

template <class D>
struct EdgeSeqs {
  using H = Hoist<D>;
  static constexpr int KE = H::KE;
  using Fwd = Seq<LY<KE, D::EH1>, LY<D::EH1, D::EH2>, LY<D::EH2, D::DE>,        // edge_update (.0: e | att columns)
                  LY<D::DE, D::MH>, LY<D::MH, D::DM>,                             // create_future_msgs (.0: e' columns)
                  LY<D::DE, D::MH>, LY<D::MH, D::DM>>;                            // create_past_msgs
  // transposed, data-gradient order
  using Bwd = Seq<LY<D::DM, D::MH>, LY<D::MH, D::DE>,                            // past.2^T, past.0[e']^T
                  LY<D::DM, D::MH>, LY<D::MH, D::DE>,                            // future.2^T, future.0[e']^T
                  LY<D::DE, D::EH2>, LY<D::EH2, D::EH1>, LY<D::EH1, KE>>;        // edge_update.4^T / .2^T / .0[e | att]^T
  using BwdNoMsg = Seq<LY<D::DE, D::EH2>, LY<D::EH2, D::EH1>, LY<D::EH1, KE>>;
  // edge_update only (round 6, B3D_FLAG_SKIP_DEAD_LAST_MESSAGES): the last layer's message stacks and node update feed nothing
  // (forward returns edge_classifier(edge_attr): clr_att_gnn.py:188)
  using FwdNoMsg = Seq<LY<KE, D::EH1>, LY<D::EH1, D::EH2>, LY<D::EH2, D::DE>>;
};

// ---- forward -----------------------------------------------------------------------------------------------------------------
// loads / stores in front of the first chunk of layers 1 .. 6, per wavefront (kRB row blocks of 16 rows: one 16-byte access per
// 16-feature block and row block)
template <class D, bool TRAIN, bool MSGS = true>
struct FwdHooks {
  using S = typename std::conditional<MSGS, typename EdgeSeqs<D>::Fwd, typename EdgeSeqs<D>::FwdNoMsg>::type;
  __host__ __device__ static constexpr int before(int ci) {
    constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, EB = D::DE / 16, MHB = D::MH / 16, DMB = D::DM / 16, SV = (TRAIN && !(B3D_ES_ABL & 32)) ? 1 : 0;
    if (!MSGS) return kRB * (ci == S::first_chunk(1) ? SV * (H1B + 1) : ci == S::first_chunk(2) ? SV * (H2B + 1) : 0);    // no message rows to gather
    return kRB * (ci == S::first_chunk(1) ? SV * (H1B + 1) + MHB    // sH1 + its mask store, T[dst] future rows
                : ci == S::first_chunk(2) ? SV * (H2B + 1) + MHB    // sH2 + mask store, T[src] past rows
                : ci == S::first_chunk(3) ? EB                      // e' store
                : ci == S::first_chunk(4) ? SV * (MHB + 1)          // sF1 + mask store
                : ci == S::first_chunk(5) ? DMB                     // fut store
                : ci == S::first_chunk(6) ? SV * (MHB + 1) : 0);    // sP1 + mask store
  }
};

template <class D, bool TRAIN_, bool MSGS = true>
__global__ __launch_bounds__(kWaves * 64, kWgPerCu) void edge_fwd_kernel(const EdgeFwdHArgs a) {
  constexpr bool TRAIN = TRAIN_ && !(B3D_ES_ABL & 32);        // (timing ablation 32: the training forward without its saved activations)
  extern __shared__ __attribute__((aligned(16))) char es_smem[];
  using H = Hoist<D>;
  using S = typename FwdHooks<D, TRAIN_, MSGS>::S;
  static_assert(D::DA > 0, "camera+LiDAR+radar widths (e | att columns)");
  constexpr int EB = D::DE / 16, AB = D::DA / 16, H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  // (read in front of the stream start, whose vmcnt(0) covers it: a load at the point of use would sit exposed at the tile's end)
  const bool past_runs = a.dst_unsorted != nullptr && __builtin_amdgcn_readfirstlane(*a.dst_unsorted) == 0;
  Ring<S, FwdHooks<D, TRAIN_, MSGS>> ring;
  ring.init(a.wpack, es_smem);
  ring.start();
  const int lane = threadIdx.x & 63;
  const int ntiles = (a.E + kTileRows - 1) / kTileRows;
  StepState st;
  st.base = ring.template slot_addr<0>();
  frag_load2(st.base + lane * 16, st.cur0, st.cur1);       // chunk 0 is complete (Ring::start)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    unsigned row[kRB], rc[kRB], s[kRB], d[kRB];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      row[rb] = (unsigned)tile * (unsigned)kTileRows + (ring.wave * kRB + rb) * 16 + (lane & 15);
      rc[rb] = row[rb] < (unsigned)a.E ? row[rb] : (unsigned)a.E - 1u;         // rows past the end compute on the last edge
      s[rb] = (unsigned)a.src[rc[rb]]; d[rb] = (unsigned)a.dst[rc[rb]];
    }
    v4f ein[kRB][EB + AB];
    {
      v4f e0[kRB][EB], a0[kRB][AB];
      load_rows<EB>(a.e_in, rc, D::DE, 0, e0);
      load_rows<AB>(a.a_in, rc, D::DA, 0, a0);
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb) {
#pragma unroll
        for (int b = 0; b < EB; ++b) ein[rb][b] = e0[rb][b];
#pragma unroll
        for (int b = 0; b < AB; ++b) ein[rb][EB + b] = a0[rb][b];
      }
    }
    v4f h1[kRB][H1B];
    {
      v4f tb[kRB][H1B];
      load_rows<H1B>(a.T, d, H::TW, H::OA, h1);
      load_rows<H1B>(a.T, s, H::TW, H::OB, tb);
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int b = 0; b < H1B; ++b) h1[rb][b] += tb[rb][b];
    }
    // ---- edge_update ----
    // (the loads / stores of a layer boundary travel as the next layer's hook: issued in front of it, or -- kLate -- inside its first step)
    {
      Bf3 x0[kRB][(EB + AB) / 2];
      split_blocks<EB + AB>(ein, x0);
      layer<S, 0, true, false, true>(ring, more, st, x0, h1);
    }
    v4f fi[kRB][MHB];
    v4f h2[kRB][H2B];
    {
      Bf3 x1[kRB][H1B / 2];
      split_blocks<H1B>(h1, x1);
      layer<S, 1, true, true, false>(ring, more, st, x1, h2, [&]() {
        if constexpr (TRAIN) { store_rows<H1B>(a.sH1, row, D::EH1, h1); store_masks<H1B, 0>(a.rmask, row, h1); }
        if constexpr (MSGS) load_rows<MHB>(a.T, d, H::TW, H::OF, fi);
      });
    }
    v4f pi[kRB][MHB];
    v4f en[kRB][EB];
    {
      Bf3 x2[kRB][H2B / 2];
      split_blocks<H2B>(h2, x2);
      layer<S, 2, false, true, false>(ring, more, st, x2, en, [&]() {
        if constexpr (TRAIN) { store_rows<H2B>(a.sH2, row, D::EH2, h2); store_masks<H2B, 2>(a.rmask, row, h2); }
        if constexpr (MSGS) load_rows<MHB>(a.T, s, H::TW, H::OP, pi);
      });
    }
    if constexpr (!MSGS) {
      store_rows<EB>(a.e_out, row, D::DE, en);                 // (behind the last rendezvous: drained at the kernel's end)
    } else {
    Bf3 xe[kRB][EB / 2];
    split_blocks<EB>(en, xe);
    // ---- create_future_msgs ----
    layer<S, 3, true, false, true>(ring, more, st, xe, fi, [&]() { store_rows<EB>(a.e_out, row, D::DE, en); });
    v4f mo[kRB][DMB];
    {
      Bf3 x4[kRB][MHB / 2];
      split_blocks<MHB>(fi, x4);
      layer<S, 4, false, true, false>(ring, more, st, x4, mo, [&]() {
        if constexpr (TRAIN) { store_rows<MHB>(a.sF1, row, D::MH, fi); store_masks<MHB, 0>(a.rmask2, row, fi); }
      });
    }
    // ---- create_past_msgs ----
    layer<S, 5, true, false, true>(ring, more, st, xe, pi, [&]() { store_rows<DMB>(a.fut, row, D::DM, mo); });
    {
      v4f mp[kRB][DMB];
      Bf3 x6[kRB][MHB / 2];
      split_blocks<MHB>(pi, x6);
      layer<S, 6, false, true, false>(ring, more, st, x6, mp, [&]() {
        if constexpr (TRAIN) { store_rows<MHB>(a.sP1, row, D::MH, pi); store_masks<MHB, 2>(a.rmask2, row, pi); }
      });
      // `past` is summed per destination (clr_att_gnn.py:293-294,336-344): with the edges grouped by destination the run sums are
      // made here and only a run's last row is kept -- the others land in the dump rows (every lane stores either way: the number of
      // vector-memory instructions between two rendezvous stays what the hook table says)
      if (past_runs) {
        unsigned prow[kRB];
#pragma unroll
        for (int rb = 0; rb < kRB; ++rb) {
          const int key = row[rb] < (unsigned)a.E ? (int)d[rb] : -2 - (int)(lane & 15);     // padding rows join no run
          const bool whole = run_sums<DMB>(mp[rb], key);
          prow[rb] = whole ? row[rb] : a.past_dump0 + (row[rb] & (unsigned)(kPastDumpRows - 1));
        }
        store_rows<DMB>(a.past, prow, D::DM, mp);
      } else {
        store_rows<DMB>(a.past, row, D::DM, mp);
      }
    }
    }   // MSGS
  }
}

// ---- backward ----------------------------------------------------------------------------------------------------------------
template <class D, bool MSGS>
struct BwdHooks {
  using S = typename std::conditional<MSGS, typename EdgeSeqs<D>::Bwd, typename EdgeSeqs<D>::BwdNoMsg>::type;
  __host__ __device__ static constexpr int before(int ci) {
    constexpr int H1B = D::EH1 / 16, H2B = D::EH2 / 16, EB = D::DE / 16, AB = D::DA / 16, MHB = D::MH / 16;
    if (MSGS)
      return kRB * (ci == S::first_chunk(1) ? MHB + D::DM / 16        // GdP1 store, dM[src] load
                  : ci == S::first_chunk(3) ? MHB + 1                 // GdF1 store, mask plane A load
                  : ci == S::first_chunk(4) ? EB                      // Gde store
                  : ci == S::first_chunk(5) ? H2B                     // GdH2 store
                  : ci == S::first_chunk(6) ? H1B + AB : 0);          // GdH1 store, running d att load
    return kRB * (ci == S::first_chunk(1) ? H2B : ci == S::first_chunk(2) ? H1B + AB : 0);    // GdH2 store; GdH1 store + d att load
  }
};

// Data gradient of the edge phase without the node columns of the three first layers (those are contracted per node from
// the segment sums of GdH1 / GdF1 / GdP1: node_listsum_kernel + node_bwd_g_kernel).
template <class D, bool MSGS>
__global__ __launch_bounds__(kWaves * 64, kWgPerCu) void edge_bwd_kernel(const EdgeBwdHArgs a) {
  extern __shared__ __attribute__((aligned(16))) char es_smem[];
  using S = typename BwdHooks<D, MSGS>::S;
  static_assert(D::DA > 0, "camera+LiDAR+radar widths (e | att columns)");
  constexpr int L0 = MSGS ? 4 : 0;
  constexpr int EB = D::DE / 16, AB = D::DA / 16, H1B = D::EH1 / 16, H2B = D::EH2 / 16, MHB = D::MH / 16, DMB = D::DM / 16;
  Ring<S, BwdHooks<D, MSGS>> ring;
  ring.init(a.wpack, es_smem);
  ring.start();
  const int lane = threadIdx.x & 63;
  const int ntiles = (a.E + kTileRows - 1) / kTileRows;
  StepState st;
  st.base = ring.template slot_addr<0>();
  frag_load2(st.base + lane * 16, st.cur0, st.cur1);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + (int)gridDim.x < ntiles;
    unsigned row[kRB], rc[kRB];
#pragma unroll
    for (int rb = 0; rb < kRB; ++rb) {
      row[rb] = (unsigned)tile * (unsigned)kTileRows + (ring.wave * kRB + rb) * 16 + (lane & 15);
      rc[rb] = row[rb] < (unsigned)a.E ? row[rb] : (unsigned)a.E - 1u;
    }
    v4f de[kRB][EB];
    load_rows<EB>(a.de_out, rc, D::DE, 0, de);
    u4v mka[kRB], mkb[kRB];                                    // the forward's ReLU masks of this lane's values (128 bytes per edge)
    if constexpr (MSGS) load_mask_plane(a.rmask2, rc, mkb);    // sF1 | sP1
    else load_mask_plane(a.rmask, rc, mka);                    // sH1 | sH2 (MSGS: behind the message layers)
    v4f d2[kRB][H2B], d1[kRB][H1B], dein[kRB][EB + AB];
    v4f prev[kRB][AB];
    if constexpr (MSGS) {
      unsigned s[kRB], d[kRB];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb) { s[rb] = (unsigned)a.src[rc[rb]]; d[rb] = (unsigned)a.dst[rc[rb]]; }
      v4f dmp[kRB][DMB], dmf[kRB][DMB];
      load_rows<DMB>(a.dM, d, 2 * D::DM, 0, dmp);             // past messages were summed at dst
      v4f dh[kRB][MHB], dee[kRB][EB], dh2[kRB][MHB];
      {
        Bf3 x0[kRB][DMB / 2];
        split_blocks<DMB>(dmp, x0);
        layer<S, 0, false, false, false>(ring, more, st, x0, dh);
      }
      relu_bwd_mask<MHB, 2>(dh, mkb);
      {
        Bf3 x1[kRB][MHB / 2];
        split_blocks<MHB>(dh, x1);
        layer<S, 1, false, false, false>(ring, more, st, x1, dee, [&]() {
          store_rows<MHB>(a.GdP1, row, D::MH, dh);
          load_rows<DMB>(a.dM, s, 2 * D::DM, D::DM, dmf);     // future messages were summed at src (needed a layer from here)
        });
      }
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int b = 0; b < EB; ++b) de[rb][b] += dee[rb][b];
      {
        Bf3 x2[kRB][DMB / 2];
        split_blocks<DMB>(dmf, x2);
        layer<S, 2, false, false, false>(ring, more, st, x2, dh2);
      }
      relu_bwd_mask<MHB, 0>(dh2, mkb);
      {
        Bf3 x3[kRB][MHB / 2];
        split_blocks<MHB>(dh2, x3);
        layer<S, 3, false, false, false>(ring, more, st, x3, dee, [&]() {
          store_rows<MHB>(a.GdF1, row, D::MH, dh2);
          load_mask_plane(a.rmask, rc, mka);
        });
      }
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb)
#pragma unroll
        for (int b = 0; b < EB; ++b) de[rb][b] += dee[rb][b];
    }
    {
      Bf3 x4[kRB][EB / 2];
      split_blocks<EB>(de, x4);
      if constexpr (MSGS) {
        layer<S, L0 + 0, false, false, false>(ring, more, st, x4, d2, [&]() { store_rows<EB>(a.Gde, row, D::DE, de); });
      } else {
        store_rows<EB>(a.Gde, row, D::DE, de);                  // (in front of chunk 0: drained by the first rendezvous like the tile's loads)
        layer<S, L0 + 0, false, false, false>(ring, more, st, x4, d2);
      }
    }
    relu_bwd_mask<H2B, 2>(d2, mka);
    {
      Bf3 x5[kRB][H2B / 2];
      split_blocks<H2B>(d2, x5);
      layer<S, L0 + 1, false, false, false>(ring, more, st, x5, d1, [&]() { store_rows<H2B>(a.GdH2, row, D::EH2, d2); });
    }
    relu_bwd_mask<H1B, 0>(d1, mka);
    {
      Bf3 x6[kRB][H1B / 2];
      split_blocks<H1B>(d1, x6);
      layer<S, L0 + 2, false, false, false>(ring, more, st, x6, dein, [&]() {
        store_rows<H1B>(a.GdH1, row, D::EH1, d1);
        load_rows<AB>(a.da_acc, rc, D::DA, 0, prev);
      });
    }
    {
      v4f o[kRB][EB], da[kRB][AB];
#pragma unroll
      for (int rb = 0; rb < kRB; ++rb) {
#pragma unroll
        for (int b = 0; b < EB; ++b) o[rb][b] = dein[rb][b];
#pragma unroll
        for (int b = 0; b < AB; ++b) da[rb][b] = a.da_first ? dein[rb][EB + b] : dein[rb][EB + b] + prev[rb][b];   // (always loaded: a fixed number of loads)
      }
      store_rows<EB>(a.de_in, row, D::DE, o);
      store_rows<AB>(a.da_acc, row, D::DA, da);
    }
  }
}

}  // namespace es
}  // namespace b3d
