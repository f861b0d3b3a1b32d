// Binned-SAH BVH build producing the reference's tree and primitive order
// (CUDA-PathTracer/tracer/bvh.cpp:8-180): 12 buckets on the centroid, all
// three axes, a leaf when <=4 primitives or when any box extent is below
// 1e-4 or when no split beats N*SA(parent); nodes numbered in pre-order and
// primitives re-ordered leaf by leaf.  The reference recurses over copies of
// the 176-byte primitives; this build sorts an index array against
// precomputed boxes instead, which yields the same tree.
#include <algorithm>
#include "scene.h"

namespace caster {

BBox prim_bbox(const Prim& p) {
  BBox b;
  if (p.type == GT_TRIANGLE) {  // Triangle.h:71-79
    b.expand(p.v[0].v); b.expand(p.v[1].v); b.expand(p.v[2].v);
  } else if (p.type == GT_SPHERE) {  // sphere.h:15-18
    f3 r = mk3(p.w0, p.w0, p.w0);
    b.lo = p.p0 - r; b.hi = p.p0 + r;
  } else {  // line.h:15-25
    float m = p.w0 > p.w1 ? p.w0 : p.w1;
    f3 w = mk3(m, m, m);
    b.expand(p.p0 - w); b.expand(p.p0 + w); b.expand(p.p1 - w); b.expand(p.p1 + w);
  }
  return b;
}

namespace {

struct Builder {
  const std::vector<BBox>& boxes;
  std::vector<f3> centres;
  std::vector<int> order;    // working permutation
  std::vector<int> scratch;
  std::vector<Node>& nodes;
  std::vector<int> out_order;  // leaf-by-leaf primitive order

  static float axis(const f3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

  int bucket_of(int prim, int a, float lo, float hi) const {
    const int bucket_num = 12;
    int no = (int)((axis(centres[prim], a) - lo) / (hi - lo) * bucket_num);
    return no == bucket_num ? no - 1 : no;
  }

  void make_leaf(int node, int begin, int end) {
    nodes[node].leaf = 1;
    nodes[node].second = -1;
    if (end > begin) {
      nodes[node].start = (int)out_order.size();
      for (int i = begin; i < end; ++i) out_order.push_back(order[i]);
      nodes[node].end = (int)out_order.size() - 1;
    }
  }

  // Builds the subtree over order[begin,end) whose box is `box`.
  void split(int begin, int end, const BBox& box) {
    int node = (int)nodes.size();
    nodes.push_back(Node{box.lo, box.hi, -1, 0, -1, -1});
    int n = end - begin;
    f3 diag = box.hi - box.lo;
    if (n <= 4 || diag.x < 0.0001f || diag.y < 0.0001f || diag.z < 0.0001f) {
      make_leaf(node, begin, end);
      return;
    }

    const int bucket_num = 12;
    int best_axis = -1, best_bucket = 0;
    float best_cost = (float)n * box.area();
    for (int a = 0; a < 3; ++a) {
      BBox bb[bucket_num];
      int cnt[bucket_num] = {0};
      float lo = axis(box.lo, a), hi = axis(box.hi, a);
      for (int i = begin; i < end; ++i) {
        int no = bucket_of(order[i], a, lo, hi);
        cnt[no]++;
        bb[no].expand(boxes[order[i]]);
      }
      for (int j = 1; j < bucket_num; ++j) {
        BBox b0, b1;
        int c0 = 0, c1 = 0;
        for (int k = 0; k < j; ++k) { b0.expand(bb[k]); c0 += cnt[k]; }
        for (int k = j; k < bucket_num; ++k) { b1.expand(bb[k]); c1 += cnt[k]; }
        float sa = (c0 == 0) ? 0.f : b0.area() * (float)c0;
        float sb = (c1 == 0) ? 0.f : b1.area() * (float)c1;
        float cost = sa + sb;
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bucket = j; }
      }
    }
    if (best_axis == -1) {
      make_leaf(node, begin, end);
      return;
    }

    // Stable partition keeps each side in its incoming order, as the
    // reference's push_back into left/right vectors does.
    float lo = axis(box.lo, best_axis), hi = axis(box.hi, best_axis);
    BBox lbox, rbox;
    int nl = 0, nr = 0;
    for (int i = begin; i < end; ++i) {
      int p = order[i];
      if (bucket_of(p, best_axis, lo, hi) < best_bucket) {
        order[begin + nl++] = p;  // nl <= i - begin, so this never overtakes the read
        lbox.expand(boxes[p]);
      } else {
        scratch[nr++] = p;
        rbox.expand(boxes[p]);
      }
    }
    for (int i = 0; i < nr; ++i) order[begin + nl + i] = scratch[i];

    split(begin, begin + nl, lbox);
    nodes[node].second = (int)nodes.size();
    split(begin + nl, end, rbox);
  }
};

}  // namespace

void prim_boxes(const std::vector<Prim>& prims, std::vector<BBox>& boxes, BBox& root_box) {
  boxes.resize(prims.size());
  root_box = BBox();
  for (size_t i = 0; i < prims.size(); ++i) {
    boxes[i] = prim_bbox(prims[i]);
    root_box.expand(boxes[i]);
  }
}

void build_bvh_host(const std::vector<BBox>& boxes, const BBox& root_box, std::vector<Node>& nodes, std::vector<int>& order) {
  nodes.clear();
  order.clear();
  if (boxes.empty()) return;
  Builder b{boxes, {}, {}, {}, nodes, {}};
  b.centres.resize(boxes.size());
  b.order.resize(boxes.size());
  b.scratch.resize(boxes.size());
  for (size_t i = 0; i < boxes.size(); ++i) {
    b.centres[i] = boxes[i].centre();
    b.order[i] = (int)i;
  }
  b.out_order.reserve(boxes.size());
  nodes.reserve(boxes.size() / 2 + 16);
  b.split(0, (int)boxes.size(), root_box);
  order.swap(b.out_order);
}

void apply_prim_order(std::vector<Prim>& prims, const std::vector<int>& order) {
  std::vector<Prim> sorted;
  sorted.reserve(prims.size());
  for (int id : order) sorted.push_back(prims[id]);
  prims.swap(sorted);
}

void build_bvh(std::vector<Prim>& prims, std::vector<Node>& nodes, BBox& root_box) {
  std::vector<BBox> boxes;
  std::vector<int> order;
  prim_boxes(prims, boxes, root_box);
  build_bvh_host(boxes, root_box, nodes, order);
  apply_prim_order(prims, order);
}

}  // namespace caster
