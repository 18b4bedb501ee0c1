// Graph handle: reference edge list -> receiver-sorted CSR + sender-sorted transpose.
// Host-side counting sorts; one-time setup.  See include/gcl.h for the contract.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace gcl {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gcl

namespace gcl {
int ensure_dyn_lds(const void* func, size_t bytes) {
  if (bytes <= 64 * 1024) return GCL_OK;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  GCL_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = done[{func, dev}];
  if (have >= bytes) return GCL_OK;
  GCL_CHECK_HIP(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return GCL_OK;
}
}  // namespace gcl

extern "C" int gcl_version(void) { return GCL_VERSION; }
extern "C" const char* gcl_last_error(void) { return gcl::g_err; }

static int check_edges(const int64_t* ei, int64_t E, int32_t n, int32_t kind) {
  GCL_CHECK_ARG(n > 0, "graph: n must be positive (got %d)", n);
  GCL_CHECK_ARG(E >= 0, "graph: negative edge count");
  GCL_CHECK_ARG(E == 0 || ei != nullptr, "graph: null edge_index");
  GCL_CHECK_ARG(kind >= GCL_GRAPH_GCN && kind <= GCL_GRAPH_MEAN, "graph: unknown kind %d", kind);
  for (int64_t e = 0; e < 2 * E; ++e)
    GCL_CHECK_ARG(ei[e] >= 0 && ei[e] < n, "graph: node index %lld out of range [0,%d)", (long long)ei[e], n);
  return GCL_OK;
}

extern "C" int gcl_graph_count_edges(const int64_t* ei, int64_t E, int32_t n, int32_t kind, int64_t* e_out) {
  int rc = check_edges(ei, E, n, kind);
  if (rc) return rc;
  GCL_CHECK_ARG(e_out, "graph: null e_out");
  if (kind == GCL_GRAPH_MEAN) {
    *e_out = E;
    return GCL_OK;
  }
  int64_t kept = 0;
  for (int64_t e = 0; e < E; ++e) kept += (ei[e] != ei[E + e]);
  *e_out = kept + n;
  GCL_CHECK_ARG(*e_out < (int64_t)INT32_MAX, "graph: too many edges for int32 CSR");
  return GCL_OK;
}

// Edge list in PyG order: kept edges in input order, then loops 0..n-1 (GCN/GAT kinds).
static void pyg_order(const int64_t* ei, int64_t E, int32_t n, int32_t kind, std::vector<int64_t>& s,
                      std::vector<int64_t>& r) {
  s.clear();
  r.clear();
  for (int64_t e = 0; e < E; ++e) {
    if (kind != GCL_GRAPH_MEAN && ei[e] == ei[E + e]) continue;
    s.push_back(ei[e]);
    r.push_back(ei[E + e]);
  }
  if (kind != GCL_GRAPH_MEAN)
    for (int32_t i = 0; i < n; ++i) {
      s.push_back(i);
      r.push_back(i);
    }
}

extern "C" int gcl_graph_build_host(const int64_t* ei, int64_t E, int32_t n, int32_t kind, int32_t* rowptr,
                                    int32_t* col, float* w, int32_t* eperm, int32_t* trowptr, int32_t* tcol,
                                    float* tw, int32_t* tslot) {
  int rc = check_edges(ei, E, n, kind);
  if (rc) return rc;
  GCL_CHECK_ARG(rowptr && col && w && eperm && trowptr && tcol && tw && tslot, "graph: null output array");
  std::vector<int64_t> s, r;
  pyg_order(ei, E, n, kind, s, r);
  const int64_t Ep = (int64_t)s.size();

  std::vector<int32_t> indeg(n, 0);
  for (int64_t e = 0; e < Ep; ++e) indeg[r[e]]++;

  // Edge weights in fp32, the way PyG forms them: deg^-1/2 per node, then dis[row]*dis[col].
  std::vector<float> node_scale(n, 1.f);
  if (kind == GCL_GRAPH_GCN) {
    for (int32_t i = 0; i < n; ++i) node_scale[i] = indeg[i] > 0 ? 1.0f / sqrtf((float)indeg[i]) : 0.f;  // torch CPU pow(-0.5) == 1/sqrt
  } else if (kind == GCL_GRAPH_MEAN) {
    for (int32_t i = 0; i < n; ++i) node_scale[i] = 1.f / (float)(indeg[i] > 1 ? indeg[i] : 1);
  }
  auto edge_w = [&](int64_t e) -> float {
    if (kind == GCL_GRAPH_GCN) return node_scale[s[e]] * node_scale[r[e]];
    if (kind == GCL_GRAPH_MEAN) return node_scale[r[e]];
    return 1.f;
  };

  // forward CSR: stable counting sort by receiver
  rowptr[0] = 0;
  for (int32_t i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + indeg[i];
  std::vector<int32_t> cur(rowptr, rowptr + n);
  std::vector<int32_t> slot_of(Ep);
  for (int64_t e = 0; e < Ep; ++e) {
    int32_t p = cur[r[e]]++;
    col[p] = (int32_t)s[e];
    w[p] = edge_w(e);
    eperm[p] = (int32_t)e;
    slot_of[e] = p;
  }
  // transpose: stable counting sort by sender
  std::vector<int32_t> outdeg(n, 0);
  for (int64_t e = 0; e < Ep; ++e) outdeg[s[e]]++;
  trowptr[0] = 0;
  for (int32_t i = 0; i < n; ++i) trowptr[i + 1] = trowptr[i] + outdeg[i];
  std::vector<int32_t> tcur(trowptr, trowptr + n);
  for (int64_t e = 0; e < Ep; ++e) {
    int32_t p = tcur[s[e]]++;
    tcol[p] = (int32_t)r[e];
    tw[p] = edge_w(e);
    tslot[p] = slot_of[e];
  }
  return GCL_OK;
}

template <typename T>
static int upload(T** dst, const T* src, size_t count) {
  *dst = nullptr;
  const size_t alloc = count ? count : 1;  // empty arrays still get a valid device pointer
  hipError_t e = hipMalloc((void**)dst, alloc * sizeof(T));
  if (e != hipSuccess) {
    gcl::set_error("graph_create: hipMalloc(%zu) failed: %s", alloc * sizeof(T), hipGetErrorString(e));
    return GCL_ENOMEM;
  }
  if (count) GCL_CHECK_HIP(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return GCL_OK;
}

// Source tiles of one CSR direction (common.h, gcl_halo).  Returns false (nothing allocated) when the tiling
// does not pay: fewer than 1.6 edge reads per staged source row, or a tile whose sources cannot fit in LDS.
namespace {
struct HaloHost {
  int32_t T = 0, ntiles = 0, smax = 0;
  std::vector<int32_t> list, cnt, rec, opos;
};

bool build_halo_host(const std::vector<int32_t>& rp, const std::vector<int32_t>& cl, const std::vector<float>& ww,
                     int32_t n, int32_t T, HaloHost& h) {
  const int32_t ntiles = (int32_t)gcl::cdiv(n, T);
  if (ntiles < 2) return false;
  // a tile's image holds its own T rows at positions [0, T) (staged without any list) and then the sources
  // outside the tile (the halo list), ascending
  std::vector<int32_t> mark(n, -1);
  std::vector<std::vector<int32_t>> src(ntiles);
  int64_t edges = 0, staged = 0;
  int32_t max_h = 0;
  for (int32_t t = 0; t < ntiles; ++t) {
    auto& sv = src[t];
    const int32_t lo = t * T, hi = (t + 1) * T;
    for (int32_t i = lo; i < n && i < hi; ++i) {
      const int32_t d = rp[i + 1] - rp[i];
      if (d > gcl::kHeavy) continue;  // heavy rows read global memory in their own kernel
      edges += d;
      for (int32_t e = rp[i]; e < rp[i + 1]; ++e) {
        const int32_t j = cl[e];
        if ((j < lo || j >= hi) && mark[j] != t) {
          mark[j] = t;
          sv.push_back(j);
        }
      }
    }
    std::sort(sv.begin(), sv.end());
    staged += T + (int64_t)sv.size();
    if ((int32_t)sv.size() > max_h) max_h = (int32_t)sv.size();
  }
  if (edges * 10 < staged * 16) return false;
  const int32_t hstride = max_h > 0 ? (max_h + 7) / 8 * 8 : 8;
  const int32_t smax = T + hstride;
  if (smax > 1024) return false;
  h.T = T;
  h.ntiles = ntiles;
  h.smax = smax;
  h.list.assign((size_t)ntiles * hstride, 0);
  h.cnt.assign(ntiles, 0);
  h.rec.assign((size_t)n * gcl::kHaloRec * 2, 0);
  h.opos.assign(cl.size(), smax);
  std::vector<int32_t> posof(n, 0);
  for (int32_t t = 0; t < ntiles; ++t) {
    const auto& sv = src[t];
    const int32_t c = (int32_t)sv.size();
    const int32_t lo = t * T, hi = (t + 1) * T;
    for (int32_t k = 0; k < hstride; ++k) h.list[(size_t)t * hstride + k] = c ? sv[k < c ? k : c - 1] : lo;
    h.cnt[t] = (c + 7) / 8 * 8;
    for (int32_t k = 0; k < c; ++k) posof[sv[k]] = T + k;
    auto pos = [&](int32_t j) { return (j >= lo && j < hi) ? j - lo : posof[j]; };
    for (int32_t i = lo; i < n && i < hi; ++i) {
      const int32_t d = rp[i + 1] - rp[i];
      const bool heavy = d > gcl::kHeavy;
      int32_t* r = &h.rec[(size_t)i * gcl::kHaloRec * 2];
      for (int k = 0; k < gcl::kHaloRec; ++k) {
        const bool in = !heavy && k < d;
        float wv = in ? ww[rp[i] + k] : 0.f;
        int32_t wb;
        memcpy(&wb, &wv, 4);
        r[2 * k] = in ? pos(cl[rp[i] + k]) : smax;
        r[2 * k + 1] = wb;
      }
      if (heavy) r[2 * (gcl::kHaloRec - 1)] |= gcl::kHaloSkip;
      else if (d > gcl::kHaloRec) r[2 * (gcl::kHaloRec - 1)] |= gcl::kHaloMore;
      if (!heavy)
        for (int32_t e = rp[i]; e < rp[i + 1]; ++e) h.opos[e] = pos(cl[e]);
    }
  }
  return true;
}
}  // namespace

namespace {
// see common.h, gcl_graph::order16
std::vector<int32_t> build_order16(const std::vector<int32_t>& rp, const std::vector<int32_t>& cl, int32_t n) {
  const int32_t ng = (n + 15) / 16;
  std::vector<int64_t> key((size_t)ng);
  bool moved = false;
  for (int32_t gi = 0; gi < ng; ++gi) {
    int32_t last = -1;
    for (int32_t r = gi * 16; r < std::min(n, gi * 16 + 16); ++r)
      for (int32_t e = rp[r]; e < rp[r + 1]; ++e)
        if (cl[e] != r) last = std::max(last, cl[e] / 16);
    if (last < 0) last = gi;  // only self-loops: stays where it is
    moved |= last != gi;
    key[(size_t)gi] = (int64_t)last * ng + gi;  // after group `last`, ties in row order
  }
  std::vector<int32_t> order((size_t)ng);
  for (int32_t gi = 0; gi < ng; ++gi) order[(size_t)gi] = gi;
  if (moved) std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return key[(size_t)a] < key[(size_t)b]; });
  return order;
}
}  // namespace

extern "C" void gcl_graph_destroy(gcl_graph_t* g) {
  if (!g) return;
  for (int d = 0; d < 2; ++d)
    if (g->order16[d]) (void)hipFree(g->order16[d]);
  for (int d = 0; d < 2; ++d)
    for (int t = 0; t < 2; ++t) {
      void* hp[] = {g->halo[d][t].list, g->halo[d][t].cnt, g->halo[d][t].rec, g->halo[d][t].opos};
      for (void* p : hp)
        if (p) (void)hipFree(p);
    }
  void* ptrs[] = {g->rowptr, g->col, g->eperm, g->trowptr, g->tcol, g->tslot, g->w, g->tw,
                  g->ecol, g->tecol, g->ew, g->tew, g->heavy, g->theavy, g->teslot};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  free(g->h_edges);
  delete g;
}

extern "C" int gcl_graph_create(const int64_t* ei, int64_t E, int32_t n, int32_t kind, gcl_graph_t** out) {
  GCL_CHECK_ARG(out, "graph_create: null out");
  *out = nullptr;
  int64_t Ep = 0;
  int rc = gcl_graph_count_edges(ei, E, n, kind, &Ep);
  if (rc) return rc;
  std::vector<int32_t> rowptr(n + 1), col(Ep + 1), eperm(Ep + 1), trowptr(n + 1), tcol(Ep + 1), tslot(Ep + 1);
  std::vector<float> w(Ep + 1), tw(Ep + 1);
  rc = gcl_graph_build_host(ei, E, n, kind, rowptr.data(), col.data(), w.data(), eperm.data(), trowptr.data(),
                            tcol.data(), tw.data(), tslot.data());
  if (rc) return rc;
  gcl_graph_t* g = new gcl_graph_t();
  g->n = n;
  g->e = Ep;
  g->kind = kind;
  for (int32_t i = 0; i < n; ++i) {
    int32_t d = rowptr[i + 1] - rowptr[i], t = trowptr[i + 1] - trowptr[i];
    if (d > g->max_in_deg) g->max_in_deg = d;
    if (t > g->max_out_deg) g->max_out_deg = t;
  }
  std::vector<int64_t> s, r;
  pyg_order(ei, E, n, kind, s, r);
  g->h_edges = (int64_t*)malloc(sizeof(int64_t) * 2 * (Ep > 0 ? Ep : 1));
  if (!g->h_edges) {
    delete g;
    gcl::set_error("graph_create: host allocation failed");
    return GCL_ENOMEM;
  }
  if (Ep) {
    memcpy(g->h_edges, s.data(), sizeof(int64_t) * Ep);
    memcpy(g->h_edges + Ep, r.data(), sizeof(int64_t) * Ep);
  }
  rc = upload(&g->rowptr, rowptr.data(), n + 1);
  if (!rc) rc = upload(&g->col, col.data(), Ep);
  if (!rc) rc = upload(&g->w, w.data(), Ep);
  if (!rc) rc = upload(&g->eperm, eperm.data(), Ep);
  if (!rc) rc = upload(&g->trowptr, trowptr.data(), n + 1);
  if (!rc) rc = upload(&g->tcol, tcol.data(), Ep);
  if (!rc) rc = upload(&g->tw, tw.data(), Ep);
  if (!rc) rc = upload(&g->tslot, tslot.data(), Ep);
  // ELL prefixes (forward and transpose) + the width that covers most rows
  auto build_ell = [&](const std::vector<int32_t>& rp, const std::vector<int32_t>& cl, const std::vector<float>& ww,
                       std::vector<int32_t>& ec, std::vector<float>& ewv) -> int {
    ec.assign((size_t)n * gcl::kEll, 0);
    ewv.assign((size_t)n * gcl::kEll, 0.f);
    int64_t hist[gcl::kEll + 2] = {0};
    for (int32_t i = 0; i < n; ++i) {
      const int32_t d = rp[i + 1] - rp[i];
      hist[d > gcl::kEll ? gcl::kEll + 1 : d]++;
      for (int k = 0; k < gcl::kEll; ++k) {
        const bool in = k < d;
        ec[(size_t)i * gcl::kEll + k] = in ? cl[rp[i] + k] : i;
        ewv[(size_t)i * gcl::kEll + k] = in ? ww[rp[i] + k] : 0.f;
      }
    }
    // width with the least expected work: every row issues `width` unconditional neighbour loads,
    // rows with more edges pay their remainder plus a fixed penalty for the dependent CSR loop
    int64_t best_cost = -1;
    int width = gcl::kEll;
    for (int cand : {1, 2, 4, 8}) {
      int64_t cost = (int64_t)n * cand;
      for (int32_t i = 0; i < n; ++i) {
        const int32_t d = rp[i + 1] - rp[i];
        if (d > cand && d <= gcl::kHeavy) cost += (d - cand) + 6;
      }
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        width = cand;
      }
    }
    (void)hist;
    return width;
  };
  auto heavy_rows = [&](const std::vector<int32_t>& rp) {
    std::vector<int32_t> h;
    for (int32_t i = 0; i < n; ++i)
      if (rp[i + 1] - rp[i] > gcl::kHeavy) h.push_back(i);
    return h;
  };
  std::vector<int32_t> ec, tec;
  std::vector<float> ewv, tewv;
  g->ell_width = build_ell(rowptr, col, w, ec, ewv);
  g->tell_width = build_ell(trowptr, tcol, tw, tec, tewv);
  auto cover = [&](const std::vector<int32_t>& rp) {
    for (int cand : {2, 4}) {
      int64_t over = 0;
      for (int32_t i = 0; i < n; ++i) over += (rp[i + 1] - rp[i]) > cand;
      if (over * 50 <= n) return cand;
    }
    return 8;
  };
  g->ell_cover = cover(rowptr);
  g->tell_cover = cover(trowptr);
  if (!rc) rc = upload(&g->ecol, ec.data(), ec.size());
  if (!rc) rc = upload(&g->ew, ewv.data(), ewv.size());
  if (!rc) rc = upload(&g->tecol, tec.data(), tec.size());
  if (!rc) rc = upload(&g->tew, tewv.data(), tewv.size());
  std::vector<int32_t> tes((size_t)n * gcl::kEll, 0);
  for (int32_t i = 0; i < n; ++i)
    for (int k = 0; k < gcl::kEll && trowptr[i] + k < trowptr[i + 1]; ++k)
      tes[(size_t)i * gcl::kEll + k] = tslot[trowptr[i] + k];
  if (!rc) rc = upload(&g->teslot, tes.data(), tes.size());
  std::vector<int32_t> hv = heavy_rows(rowptr), thv = heavy_rows(trowptr);
  g->n_heavy = (int32_t)hv.size();
  g->n_theavy = (int32_t)thv.size();
  if (!rc) rc = upload(&g->heavy, hv.data(), hv.size());
  if (!rc) rc = upload(&g->theavy, thv.data(), thv.size());
  // source-tile layouts (forward and transpose, T = 64 and 32)
  // GAT graphs carry no edge weights: the second word of a TRANSPOSED edge record holds the edge's forward CSR slot
  // instead (where its attention weight and score gradient live: gat_halo_bwd_src_kernel)
  std::vector<float> tpay(tw);
  if (kind == GCL_GRAPH_GAT)
    for (int64_t e = 0; e < Ep; ++e) memcpy(&tpay[e], &tslot[e], 4);
  for (int d = 0; d < 2 && !rc; ++d)
    for (int t = 0; t < 2 && !rc; ++t) {
      HaloHost hh;
      const bool ok = d == 0 ? build_halo_host(rowptr, col, w, n, t == 0 ? 64 : 32, hh)
                             : build_halo_host(trowptr, tcol, tpay, n, t == 0 ? 64 : 32, hh);
      if (!ok) continue;
      gcl_halo& H = g->halo[d][t];
      rc = upload(&H.list, hh.list.data(), hh.list.size());
      if (!rc) rc = upload(&H.cnt, hh.cnt.data(), hh.cnt.size());
      if (!rc) rc = upload(&H.rec, hh.rec.data(), hh.rec.size());
      if (!rc) rc = upload(&H.opos, hh.opos.data(), (size_t)Ep);
      H.T = hh.T;
      H.ntiles = hh.ntiles;
      H.smax = hh.smax;
    }
  // processing order of the per-edge kernels: only where one sample's rows cannot sit in an XCD's 4 MiB L2 anyway
  // (>= 32 Ki rows: 16 MB at 128 channels) and the graph has no source-tile layout
  {
    static const int ord_env = [] { const char* e = getenv("GCL_AGG_ORDER"); return e ? atoi(e) : 1; }();
    if (!rc && ord_env && n >= 32768) {
      for (int d = 0; d < 2 && !rc; ++d) {
        if (g->halo[d][0].T) continue;
        const std::vector<int32_t> ord = d == 0 ? build_order16(rowptr, col, n) : build_order16(trowptr, tcol, n);
        rc = upload(&g->order16[d], ord.data(), ord.size());
      }
      g->n_order16 = (n + 15) / 16;
    }
  }
  if (rc) {
    gcl_graph_destroy(g);
    return rc;
  }
  *out = g;
  return GCL_OK;
}

extern "C" int32_t gcl_graph_num_nodes(const gcl_graph_t* g) { return g ? g->n : 0; }
extern "C" int64_t gcl_graph_num_edges(const gcl_graph_t* g) { return g ? g->e : 0; }
extern "C" int32_t gcl_graph_max_in_degree(const gcl_graph_t* g) { return g ? g->max_in_deg : 0; }
extern "C" const int32_t* gcl_graph_eperm_device(const gcl_graph_t* g) { return g ? g->eperm : nullptr; }
extern "C" int gcl_graph_halo_info(const gcl_graph_t* g, int32_t transpose, int32_t T, int32_t* out4) {
  GCL_CHECK_ARG(g && out4, "graph_halo_info: null argument");
  GCL_CHECK_ARG(T == 64 || T == 32, "graph_halo_info: T must be 64 or 32");
  const gcl_halo& h = g->halo[transpose ? 1 : 0][T == 64 ? 0 : 1];
  out4[0] = h.T;
  out4[1] = h.ntiles;
  out4[2] = h.smax;
  out4[3] = 0;
  return GCL_OK;
}
extern "C" int gcl_graph_export_edges(const gcl_graph_t* g, int64_t* out) {
  GCL_CHECK_ARG(g && out, "graph_export_edges: null argument");
  memcpy(out, g->h_edges, sizeof(int64_t) * 2 * g->e);
  return GCL_OK;
}
