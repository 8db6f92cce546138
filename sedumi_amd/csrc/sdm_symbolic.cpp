// sdm_symbolic.cpp -- once-per-solve integer work of the hot path (host C++): ordering, symbolic
// factorisation, work-size / cache-split tables.  Results are BIT-EXACT with the reference:
//   sdm_ordmmd      <-> ordmmdmex.c:75-139 + ordmmd.c (SPARSPAK GENMMD, Liu's multiple minimum external
//                       degree with delta = 0, maxint = 32767; ordmmd.c:86-87)
//   sdm_symfct      <-> symfctmex.c:127-272 + symfct.c sfinit_/symfct_ (Ng & Peyton: elimination tree,
//                       postorder, column counts, child reordering, maximal supernodes, supernodal structure)
//   sdm_choltmpsiz  <-> choltmpsiz.c:57-101,   sdm_cholsplit <-> cholsplit.c:59-111
// The ordering keeps the published algorithm's data structures (quotient graph stored in place in the
// adjacency array, degree lists threaded through two index arrays) because the permutation it returns is
// defined by their tie-breaking; the symbolic factorisation only has to agree on the two postorders --
// column counts, supernodes and row structures are mathematically determined and are computed here by a
// direct row-subtree / child-merge formulation.  This is graph code with no data parallelism worth a GPU
// (SURVEY.md section 7, step 6).
#include "../../include/sedumi_hip.h"
#include <algorithm>
#include <cmath>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace sdm {
void set_error(const std::string &msg);
typedef int64_t Int;

// ------------------------------------------------------------------ adjacency (getadj, ordmmdmex.c:55-66)
// 1-based adjacency structure without the diagonal; xadj has m+2 slots so that index m+1 is valid.
static void build_adjacency(Int m, const Int *jc, const Int *ir, std::vector<Int> &xadj, std::vector<Int> &adj) {
  xadj.assign(m + 2, 0);
  adj.assign(1, 0);
  adj.reserve((size_t)jc[m] + 2);
  for (Int j = 0; j < m; j++) {
    xadj[j + 1] = (Int)adj.size();
    for (Int t = jc[j]; t < jc[j + 1]; t++)
      if (ir[t] != j) adj.push_back(ir[t] + 1);
  }
  xadj[m + 1] = (Int)adj.size();
}

// =========================================================== minimum degree
class MinDegree {
 public:
  MinDegree(Int n, std::vector<Int> &xadj, std::vector<Int> &adj)
      : n_(n), xadj_(xadj), adj_(adj), head_(n + 2, 0), fwd_(n + 2, 0), bwd_(n + 2, 0), qsize_(n + 2, 1),
        list_(n + 2, 0), mark_(n + 2, 0) {}

  // returns perm (1-based: perm[k] = original node eliminated k-th), k = 1..n
  std::vector<Int> run() {
    const Int delta = 0;
    if (n_ <= 0) return std::vector<Int>(1, 0);
    seed_degree_lists();
    Int num = 1;
    // isolated nodes first
    for (Int nd = head_[1]; nd > 0;) {
      Int nxt = fwd_[nd];
      mark_[nd] = kBig; fwd_[nd] = -num; ++num;
      nd = nxt;
    }
    if (num <= n_) {
      tag_ = 1; head_[1] = 0;
      Int mdeg = 2;
      bool done = false;
      while (!done) {
        while (head_[mdeg] <= 0) ++mdeg;
        const Int limit = mdeg + delta;
        Int ehead = 0;
        for (;;) {
          Int nd = head_[mdeg];
          if (nd <= 0) { ++mdeg; if (mdeg > limit) break; continue; }
          Int nxt = fwd_[nd];
          head_[mdeg] = nxt;
          if (nxt > 0) bwd_[nxt] = -mdeg;
          fwd_[nd] = -num;
          if (num + qsize_[nd] > n_) { done = true; break; }
          bump_tag(1);
          eliminate(nd);
          num += qsize_[nd];
          list_[nd] = ehead; ehead = nd;
          if (delta < 0) break;
        }
        if (done || num > n_) break;
        update_degrees(ehead, delta, mdeg);
      }
    }
    return final_numbering();
  }

 private:
  static constexpr Int kBig = 32767;      // "maxint" of ordmmd.c:87 -- also the marker of absorbed nodes
  Int n_, tag_ = 0;
  std::vector<Int> &xadj_, &adj_;
  std::vector<Int> head_, fwd_, bwd_, qsize_, list_, mark_;

  void seed_degree_lists() {
    for (Int nd = 1; nd <= n_; nd++) {
      Int deg = xadj_[nd + 1] - xadj_[nd] + 1;
      Int f = head_[deg];
      fwd_[nd] = f; head_[deg] = nd;
      if (f > 0) bwd_[f] = nd;
      bwd_[nd] = -deg;
    }
  }
  // advance the marker tag by `by`; when it would reach maxint all live markers are reset
  void bump_tag(Int by) {
    if (tag_ + by < kBig) { tag_ += by; return; }
    tag_ = 1;
    for (Int i = 1; i <= n_; i++) if (mark_[i] < kBig) mark_[i] = 0;
  }

  // eliminate `md`: build its element (reachable set) in place and strip it from the neighbours' lists
  void eliminate(Int md) {
    mark_[md] = tag_;
    const Int istrt = xadj_[md], istop = xadj_[md + 1] - 1;
    Int elmnt = 0, rloc = istrt, rlmt = istop;
    for (Int i = istrt; i <= istop; i++) {
      Int nb = adj_[i];
      if (nb == 0) break;
      if (mark_[nb] >= tag_) continue;
      mark_[nb] = tag_;
      if (fwd_[nb] < 0) { list_[nb] = elmnt; elmnt = nb; }       // an already eliminated neighbour (element)
      else adj_[rloc++] = nb;
    }
    while (elmnt > 0) {                                           // merge the reach of the adjacent elements
      adj_[rlmt] = -elmnt;
      Int link = elmnt;
      bool follow = true;
      while (follow) {
        follow = false;
        const Int js = xadj_[link], je = xadj_[link + 1] - 1;
        for (Int j = js; j <= je; j++) {
          Int nd = adj_[j];
          if (nd < 0) { link = -nd; follow = true; break; }
          if (nd == 0) break;
          if (mark_[nd] >= tag_ || fwd_[nd] < 0) continue;
          mark_[nd] = tag_;
          while (rloc >= rlmt) { Int l2 = -adj_[rlmt]; rloc = xadj_[l2]; rlmt = xadj_[l2 + 1] - 1; }
          adj_[rloc++] = nd;
        }
      }
      elmnt = list_[elmnt];
    }
    if (rloc <= rlmt) adj_[rloc] = 0;
    // every node of the new element: leave its degree list, purge marked neighbours, attach the element
    Int link = md;
    bool follow = true;
    while (follow) {
      follow = false;
      const Int is = xadj_[link], ie = xadj_[link + 1] - 1;
      for (Int i = is; i <= ie; i++) {
        Int r = adj_[i];
        if (r < 0) { link = -r; follow = true; break; }
        if (r == 0) return;
        Int pv = bwd_[r];
        if (pv != 0 && pv != -kBig) {
          Int nx = fwd_[r];
          if (nx > 0) bwd_[nx] = pv;
          if (pv > 0) fwd_[pv] = nx;
          if (pv < 0) head_[-pv] = nx;
        }
        const Int js = xadj_[r], je = xadj_[r + 1] - 1;
        Int keep = js;
        for (Int j = js; j <= je; j++) {
          Int nb = adj_[j];
          if (nb == 0) break;
          if (mark_[nb] >= tag_) continue;
          adj_[keep++] = nb;
        }
        if (keep - js <= 0) {                                     // indistinguishable from md: absorb
          qsize_[md] += qsize_[r]; qsize_[r] = 0;
          mark_[r] = kBig; fwd_[r] = -md; bwd_[r] = -kBig;
        } else {
          fwd_[r] = keep - js + 1; bwd_[r] = 0;
          adj_[keep++] = md;
          if (keep <= je) adj_[keep] = 0;
        }
      }
    }
  }

  void insert_with_degree(Int nd, Int deg, Int &mdeg) {
    deg = deg - qsize_[nd] + 1;
    Int f = head_[deg];
    fwd_[nd] = f; bwd_[nd] = -deg;
    if (f > 0) bwd_[f] = nd;
    head_[deg] = nd;
    if (deg < mdeg) mdeg = deg;
  }

  // recompute the external degrees of the nodes touched by the elements eliminated in this pass
  void update_degrees(Int ehead, Int delta, Int &mdeg) {
    const Int mdeg0 = mdeg + delta;
    for (Int elmnt = ehead; elmnt > 0; elmnt = list_[elmnt]) {
      Int mtag = tag_ + mdeg0;
      if (mtag >= kBig) {
        tag_ = 1;
        for (Int i = 1; i <= n_; i++) if (mark_[i] < kBig) mark_[i] = 0;
        mtag = tag_ + mdeg0;
      }
      Int q2head = 0, qxhead = 0, deg0 = 0;
      {                                                           // classify the element's nodes
        Int link = elmnt; bool follow = true;
        while (follow) {
          follow = false;
          const Int is = xadj_[link], ie = xadj_[link + 1] - 1;
          for (Int i = is; i <= ie; i++) {
            Int e = adj_[i];
            if (e < 0) { link = -e; follow = true; break; }
            if (e == 0) break;
            if (qsize_[e] == 0) continue;
            deg0 += qsize_[e];
            mark_[e] = mtag;
            if (bwd_[e] != 0) continue;
            if (fwd_[e] == 2) { list_[e] = q2head; q2head = e; }
            else { list_[e] = qxhead; qxhead = e; }
          }
        }
      }
      // nodes adjacent to exactly two elements: degree via the other element, with mass elimination
      for (Int e = q2head; e > 0; e = list_[e]) {
        if (bwd_[e] != 0) continue;
        ++tag_;
        Int deg = deg0;
        Int nb = adj_[xadj_[e]];
        if (nb == elmnt) nb = adj_[xadj_[e] + 1];
        if (fwd_[nb] >= 0) {
          deg += qsize_[nb];
        } else {
          Int link = nb; bool follow = true, stop = false;
          while (follow && !stop) {
            follow = false;
            const Int is = xadj_[link], ie = xadj_[link + 1] - 1;
            for (Int i = is; i <= ie; i++) {
              Int nd = adj_[i];
              if (nd == e) continue;
              if (nd < 0) { link = -nd; follow = true; break; }
              if (nd == 0) { stop = true; break; }
              if (qsize_[nd] == 0) continue;
              if (mark_[nd] < tag_) { mark_[nd] = tag_; deg += qsize_[nd]; continue; }
              if (bwd_[nd] != 0) continue;
              if (fwd_[nd] == 2) {                                // same two elements: merge into e
                qsize_[e] += qsize_[nd]; qsize_[nd] = 0;
                mark_[nd] = kBig; fwd_[nd] = -e; bwd_[nd] = -kBig;
              } else if (bwd_[nd] == 0) {
                bwd_[nd] = -kBig;
              }
            }
          }
        }
        insert_with_degree(e, deg, mdeg);
      }
      // all other nodes: full scan of neighbours and of the elements they belong to
      for (Int e = qxhead; e > 0; e = list_[e]) {
        if (bwd_[e] != 0) continue;
        ++tag_;
        Int deg = deg0;
        const Int is = xadj_[e], ie = xadj_[e + 1] - 1;
        for (Int i = is; i <= ie; i++) {
          Int nb = adj_[i];
          if (nb == 0) break;
          if (mark_[nb] >= tag_) continue;
          mark_[nb] = tag_;
          if (fwd_[nb] >= 0) { deg += qsize_[nb]; continue; }
          Int link = nb; bool follow = true;
          while (follow) {
            follow = false;
            const Int js = xadj_[link], je = xadj_[link + 1] - 1;
            for (Int j = js; j <= je; j++) {
              Int nd = adj_[j];
              if (nd < 0) { link = -nd; follow = true; break; }
              if (nd == 0) break;
              if (mark_[nd] >= tag_) continue;
              mark_[nd] = tag_; deg += qsize_[nd];
            }
          }
        }
        insert_with_degree(e, deg, mdeg);
      }
      tag_ = mtag;
    }
  }

  // turn the elimination numbers (negated in fwd_) and the absorption forest into perm / invp (mmdnum)
  std::vector<Int> final_numbering() {
    std::vector<Int> &invp = fwd_, &perm = bwd_;
    for (Int nd = 1; nd <= n_; nd++) perm[nd] = qsize_[nd] <= 0 ? invp[nd] : -invp[nd];
    for (Int nd = 1; nd <= n_; nd++) {
      if (perm[nd] > 0) continue;
      Int father = nd;
      while (perm[father] <= 0) father = -perm[father];
      const Int root = father;
      const Int num = perm[root] + 1;
      invp[nd] = -num; perm[root] = num;
      father = nd;
      for (;;) {                                                  // path compression towards the root
        Int nextf = -perm[father];
        if (nextf <= 0) break;
        perm[father] = -root;
        father = nextf;
      }
    }
    std::vector<Int> out(n_ + 1, 0);
    for (Int nd = 1; nd <= n_; nd++) out[-invp[nd]] = nd;
    return out;
  }
};

// ===================================================== symbolic factorisation
struct Forest {
  std::vector<Int> parent;          // 1-based, 0 = root
};

// elimination tree of A(perm,perm) with path compression (Liu); all arrays 1-based
static std::vector<Int> elimination_tree(Int n, const std::vector<Int> &xadj, const std::vector<Int> &adj,
                                         const std::vector<Int> &perm, const std::vector<Int> &invp) {
  std::vector<Int> parent(n + 1, 0), anc(n + 1, 0);
  for (Int i = 1; i <= n; i++) {
    const Int node = perm[i];
    for (Int t = xadj[node]; t < xadj[node + 1]; t++) {
      Int k = invp[adj[t]];
      if (k >= i) continue;
      while (anc[k] != i) {
        if (anc[k] > 0) { Int nxt = anc[k]; anc[k] = i; k = nxt; }
        else { parent[k] = i; anc[k] = i; break; }
      }
    }
  }
  return parent;
}

// Postorder used by etpost_/epost2_: the forest's roots are visited from the highest-numbered one down,
// children in the order of the (first-son, brother) lists given.  Returns newlabel[old] (1-based).
static std::vector<Int> postorder(Int n, const std::vector<Int> &fson, const std::vector<Int> &brothr) {
  std::vector<Int> label(n + 1, 0), stack;
  stack.reserve(n);
  Int num = 0, node = n;
  for (;;) {
    while (node > 0) { stack.push_back(node); node = fson[node]; }
    if (stack.empty()) break;
    node = stack.back(); stack.pop_back();
    label[node] = ++num;
    node = brothr[node];
  }
  return label;
}

// first-son / brother lists: children in increasing order, roots chained from n downwards (betree_)
static void sibling_lists(Int n, const std::vector<Int> &parent, std::vector<Int> &fson, std::vector<Int> &brothr) {
  fson.assign(n + 1, 0); brothr.assign(n + 1, 0);
  Int lroot = n;
  for (Int node = n - 1; node >= 1; node--) {
    Int p = parent[node];
    if (p <= 0 || p == node) { brothr[lroot] = node; lroot = node; }
    else { brothr[node] = fson[p]; fson[p] = node; }
  }
  if (n >= 1) brothr[lroot] = 0;
}

// the same with the child order of btree2_: a child whose column count is >= that of the current LAST son goes
// to the front, otherwise to the back of its parent's list
static void sibling_lists_by_count(Int n, const std::vector<Int> &parent, const std::vector<Int> &colcnt,
                                   std::vector<Int> &fson, std::vector<Int> &brothr) {
  fson.assign(n + 1, 0); brothr.assign(n + 1, 0);
  std::vector<Int> lson(n + 1, 0);
  Int lroot = n;
  for (Int node = n - 1; node >= 1; node--) {
    Int p = parent[node];
    if (p <= 0 || p == node) { brothr[lroot] = node; lroot = node; continue; }
    Int last = lson[p];
    if (last == 0) { fson[p] = node; lson[p] = node; }
    else if (colcnt[node] >= colcnt[last]) { brothr[node] = fson[p]; fson[p] = node; }
    else { brothr[last] = node; lson[p] = node; }
  }
  if (n >= 1) brothr[lroot] = 0;
}

static void relabel(Int n, const std::vector<Int> &label, std::vector<Int> &parent, std::vector<Int> &perm,
                    std::vector<Int> &invp) {
  std::vector<Int> np(n + 1, 0);
  for (Int v = 1; v <= n; v++) np[label[v]] = parent[v] > 0 ? label[parent[v]] : 0;
  parent.swap(np);
  for (Int i = 1; i <= n; i++) invp[i] = label[invp[i]];          // invinv_
  for (Int i = 1; i <= n; i++) perm[invp[i]] = i;
}

// column counts of L for the (postordered) matrix: row-subtree traversal with a visit mark per row
static std::vector<Int> column_counts(Int n, const std::vector<Int> &xadj, const std::vector<Int> &adj,
                                      const std::vector<Int> &perm, const std::vector<Int> &invp,
                                      const std::vector<Int> &parent) {
  std::vector<Int> cnt(n + 1, 1), seen(n + 1, 0);
  for (Int i = 1; i <= n; i++) {
    seen[i] = i;
    const Int node = perm[i];
    for (Int t = xadj[node]; t < xadj[node + 1]; t++) {
      Int k = invp[adj[t]];
      while (k < i && seen[k] != i) { seen[k] = i; cnt[k]++; k = parent[k]; }
    }
  }
  return cnt;
}

struct Symbolic {
  std::vector<Int> perm, xsuper, Ljc, Lir;     // 0-based outputs
  Int nsuper = 0;
};

static Symbolic symbolic_factor(Int n, const Int *Xjc, const Int *Xir, const Int *perm_in) {
  Symbolic S;
  std::vector<Int> xadj, adj;
  build_adjacency(n, Xjc, Xir, xadj, adj);
  std::vector<Int> perm(n + 1), invp(n + 1);
  for (Int i = 0; i < n; i++) {
    if (perm_in[i] < 0 || perm_in[i] >= n) throw std::runtime_error("symfct: perm entry out of range");
    perm[i + 1] = perm_in[i] + 1; invp[perm_in[i] + 1] = i + 1;
  }
  if ((Int)adj.size() - 1 == 0) {                                  // diagonal matrix (symfct.c:117-139)
    S.perm.assign(perm_in, perm_in + n);
    S.nsuper = n; S.xsuper.resize(n + 1); S.Ljc.resize(n + 1); S.Lir.resize(n);
    for (Int j = 0; j <= n; j++) { S.xsuper[j] = j; S.Ljc[j] = j; }
    for (Int j = 0; j < n; j++) S.Lir[j] = j;
    return S;
  }
  // etordr_: etree + postorder
  std::vector<Int> parent = elimination_tree(n, xadj, adj, perm, invp), fson, brothr;
  sibling_lists(n, parent, fson, brothr);
  relabel(n, postorder(n, fson, brothr), parent, perm, invp);
  // fcnthn_ + chordr_: column counts, children reordered by count, second postorder
  std::vector<Int> cnt = column_counts(n, xadj, adj, perm, invp, parent);
  sibling_lists_by_count(n, parent, cnt, fson, brothr);
  {
    std::vector<Int> label = postorder(n, fson, brothr), nc(n + 1, 0);
    for (Int v = 1; v <= n; v++) nc[label[v]] = cnt[v];
    cnt.swap(nc);
    relabel(n, label, parent, perm, invp);
  }
  // fsup1_/fsup2_: column k joins the supernode of k-1 iff parent(k-1)=k and cnt(k-1)=cnt(k)+1
  std::vector<Int> xs(1, 1), snode(n + 1, 0);
  snode[1] = 1;
  for (Int k = 2; k <= n; k++) {
    if (!(parent[k - 1] == k && cnt[k - 1] == cnt[k] + 1)) xs.push_back(k);
    snode[k] = (Int)xs.size();
  }
  const Int nsuper = (Int)xs.size();
  xs.push_back(n + 1);
  // symfct_: row structure of each supernode's first column = {first} U adj(first) U children's structures
  std::vector<std::vector<Int>> rows(nsuper + 1), kids(nsuper + 1);
  std::vector<Int> mark(n + 1, 0);
  for (Int s = 1; s <= nsuper; s++) {
    const Int f = xs[s - 1], width = xs[s] - f;
    std::vector<Int> &r = rows[s];
    r.push_back(f); mark[f] = s;
    for (Int c : kids[s])
      for (size_t t = (size_t)(xs[c] - xs[c - 1]); t < rows[c].size(); t++) {
        Int i = rows[c][t];
        if (mark[i] != s) { mark[i] = s; r.push_back(i); }
      }
    const Int node = perm[f];
    for (Int t = xadj[node]; t < xadj[node + 1]; t++) {
      Int i = invp[adj[t]];
      if (i > f && mark[i] != s) { mark[i] = s; r.push_back(i); }
    }
    std::sort(r.begin(), r.end());
    if ((Int)r.size() != cnt[f]) throw std::runtime_error("symfct: structure / column count mismatch");
    if ((Int)r.size() > width) kids[snode[r[width]]].push_back(s);
  }
  // expandsub (symfctmex.c:90-120): every column carries its full row list
  S.nsuper = nsuper;
  S.perm.resize(n); S.xsuper.resize(nsuper + 1); S.Ljc.assign(n + 1, 0);
  for (Int i = 1; i <= n; i++) S.perm[i - 1] = perm[i] - 1;
  for (Int s = 0; s <= nsuper; s++) S.xsuper[s] = xs[s] - 1;
  for (Int j = 1; j <= n; j++) S.Ljc[j] = S.Ljc[j - 1] + cnt[j];
  S.Lir.resize(S.Ljc[n]);
  for (Int s = 1; s <= nsuper; s++) {
    const Int f = xs[s - 1];
    for (Int j = f; j < xs[s]; j++) {
      Int pos = S.Ljc[j - 1];
      for (size_t t = (size_t)(j - f); t < rows[s].size(); t++) S.Lir[pos++] = rows[s][t] - 1;
    }
  }
  return S;
}

}  // namespace sdm
using namespace sdm;

extern "C" {

int sdm_ordmmd(sdm_int m, const sdm_int *Xjc, const sdm_int *Xir, sdm_int *perm) {
  try {
    std::vector<Int> xadj, adj;
    build_adjacency(m, Xjc, Xir, xadj, adj);
    adj.push_back(0);
    MinDegree md(m, xadj, adj);
    std::vector<Int> p = md.run();
    for (Int k = 1; k <= m; k++) perm[k - 1] = p[k] - 1;
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}

int sdm_symfct(sdm_int m, const sdm_int *Xjc, const sdm_int *Xir, const sdm_int *perm_in, sdm_int *perm_out,
               sdm_int *nsuper, sdm_int *xsuper, sdm_int *nnzl, sdm_int *Ljc, sdm_int *Lir) {
  try {
    Symbolic S = symbolic_factor(m, Xjc, Xir, perm_in);
    if (nsuper) *nsuper = S.nsuper;
    if (nnzl) *nnzl = S.Ljc[m];
    if (perm_out) std::copy(S.perm.begin(), S.perm.end(), perm_out);
    if (xsuper) std::copy(S.xsuper.begin(), S.xsuper.end(), xsuper);
    if (Ljc) std::copy(S.Ljc.begin(), S.Ljc.end(), Ljc);
    if (Lir) std::copy(S.Lir.begin(), S.Lir.end(), Lir);
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}

// scratch bound of the reference's precorrect (choltmpsiz.c:57-101): max over (affecting supernode k,
// affected supernode j) of mk*q - q(q-1)/2.  Kept for interface parity: the multifrontal factor does not use it.
int sdm_choltmpsiz(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, sdm_int nsuper, const sdm_int *xsuper,
                   sdm_int *tmpsiz) {
  try {
    std::vector<Int> snode(m > 0 ? m : 1);
    for (Int s = 0; s < nsuper; s++) for (Int j = xsuper[s]; j < xsuper[s + 1]; j++) snode[j] = s;
    Int best = 0;
    for (Int s = 0; s < nsuper; s++) {
      const Int k = xsuper[s];
      Int t = Ljc[k] + (xsuper[s + 1] - k);
      const Int end = Ljc[k + 1];
      Int mk = end - t;
      while (t < end && mk * (mk + 1) / 2 > best) {
        const Int nextj = xsuper[snode[Lir[t]] + 1];
        Int q = 0;
        while (t < end && Lir[t] < nextj) { q++; t++; }
        best = std::max(best, mk * q - q * (q - 1) / 2);
        mk -= q;
      }
    }
    *tmpsiz = best;
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}

// cache groups of cholsplit.c:59-111 (vestigial in the reference too: blkchol never reads L.split).
// cachsz in KB as passed by symbchol.m:66,83; cachesiz = floor(0.9 * 128 * cachsz) doubles.
int sdm_cholsplit(sdm_int m, const sdm_int *Ljc, sdm_int nsuper, const sdm_int *xsuper, double cachsz,
                  sdm_int *split) {
  try {
    const Int cache = (Int)std::floor(0.9 * (1024 / sizeof(double)) * cachsz);
    std::fill(split, split + m, 0);
    Int k = 0;
    for (Int s = 0; s < nsuper; s++) {
      Int mk = Ljc[k + 1] - Ljc[k];
      Int used = 2 * mk;
      const Int nextk = xsuper[s + 1];
      Int j = k;
      if (used > cache) {
        k = j + (used - cache) / 2;
        if (k >= nextk) k = nextk;
        else { mk -= k - j; used = 2 * mk; }
        split[j] = k - j;
        j = k;
      } else { k++; mk--; }
      for (; k < nextk; k++, mk--) {
        if (used + mk < cache) used += mk;
        else { split[j] = k - j; j = k; used = 2 * mk; }
      }
      if (j < nextk) split[j] = nextk - j;
    }
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}

// [perm, dz] = incorder(At, Ajc1, ifirst)                                    incorder.c:140-209
// Greedy ordering of the columns of At(first:end, :): step k takes the remaining column with the fewest subscripts not
// yet covered (ties: the one standing EARLIEST in the current perm array, which the swaps of earlier steps have
// rearranged -- incorder.c:172-182), dz(:,k) = its newly covered subscripts in the order they appear in the column.
// The reference rescans all remaining columns at every step (O(m^2)); here the remaining columns sit in an ordered set
// keyed on (remaining length, position in perm), updated when a covered subscript shortens a column and when a swap
// moves one: O((nnz + m) log m), same output bit for bit.
// Ajc1 = start offsets of the rows >= first per column (NULL: column starts, first = 0).  dzir needs room for
// min(N - first, number of nonzeros in range) entries.
int sdm_incorder(sdm_int N, sdm_int m, const sdm_int *Atjc, const sdm_int *Atir, const sdm_int *Ajc1, sdm_int first,
                 sdm_int *perm, sdm_int *dzjc, sdm_int *dzir) {
  try {
    if (first < 0 || first > N) throw std::runtime_error("incorder: first subscript out of range");
    const Int lenud = N - first;
    std::vector<Int> beg(m), len(m), pos(m);
    for (Int j = 0; j < m; j++) {
      beg[j] = Ajc1 ? Ajc1[j] : Atjc[j];
      if (beg[j] < Atjc[j] || beg[j] > Atjc[j + 1]) throw std::runtime_error("incorder: Ajc1 outside its column");
      len[j] = Atjc[j + 1] - beg[j];
      perm[j] = j; pos[j] = j;
    }
    // A = At(first:end,:)' : for every subscript the columns that contain it (spPartTransp, incorder.c:78-118)
    std::vector<Int> Ajc((size_t)lenud + 1, 0), Air;
    for (Int j = 0; j < m; j++)
      for (Int t = beg[j]; t < Atjc[j + 1]; t++) {
        if (Atir[t] < first || Atir[t] >= N) throw std::runtime_error("incorder: subscript outside first:end");
        Ajc[(size_t)(Atir[t] - first) + 1]++;
      }
    for (Int i = 0; i < lenud; i++) Ajc[i + 1] += Ajc[i];
    Air.resize((size_t)Ajc[lenud]);
    { std::vector<Int> nxt(Ajc.begin(), Ajc.end() - 1);
      for (Int j = 0; j < m; j++) for (Int t = beg[j]; t < Atjc[j + 1]; t++) Air[(size_t)nxt[Atir[t] - first]++] = j; }
    std::set<std::pair<Int, Int>> rem;                          // (remaining length, position in perm) of the columns not yet taken
    for (Int j = 0; j < m; j++) rem.insert({len[j], j});
    std::vector<char> covered((size_t)std::max<Int>(lenud, 1), 0);
    dzjc[0] = 0;
    for (Int k = 0; k < m; k++) {
      const std::pair<Int, Int> best = *rem.begin();            // fewest uncovered subscripts, earliest position
      rem.erase(rem.begin());
      const Int kmin = best.second, permk = perm[kmin];
      if (kmin != k) {                                          // the swap of incorder.c:180-182
        const Int other = perm[k];
        rem.erase({len[other], k});
        perm[kmin] = other; pos[other] = kmin;
        rem.insert({len[other], kmin});
        perm[k] = permk; pos[permk] = k;
      }
      Int jnz = dzjc[k];
      for (Int t = beg[permk]; t < Atjc[permk + 1]; t++) {
        const Int i = Atir[t] - first;
        if (!covered[(size_t)i]) { covered[(size_t)i] = 1; dzir[jnz++] = Atir[t]; }
      }
      dzjc[k + 1] = jnz;
      for (Int q = dzjc[k]; q < jnz; q++) {
        const Int i = dzir[q] - first;
        for (Int t = Ajc[(size_t)i]; t < Ajc[(size_t)i + 1]; t++) {
          const Int j = Air[(size_t)t];
          if (pos[j] > k) { rem.erase({len[j], pos[j]}); len[j]--; rem.insert({len[j], pos[j]}); }
          else len[j]--;
        }
      }
    }
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}

}  // extern "C"
