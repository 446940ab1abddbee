// c_abi_host.cpp -- a host for libeinx_hip.so that knows nothing about Python or torch: plain HIP memory, the C ABI of
// include/einx.h, one event-image pair batch through two SuperPoint-shaped extractors (handle-level einx_extract) and
// the mutual-nearest-neighbour matcher.  It is what a C / C++ / cgo / JNI binding of the library does, and it doubles as a
// check that no torch type or allocator is needed below the boundary.
//
//   hipcc -O2 --offload-arch=gfx950 examples/c_abi_host.cpp -Iinclude -Lei-nexus_official_amd -leinx_hip \
//         -Wl,-rpath,$PWD/ei-nexus_official_amd -o examples/c_abi_host && examples/c_abi_host [B [IN.bin OUT.bin [lg]]]
//
// Weights are seeded pseudo-random (the network shape is SuperPointv1's / VGGExtractor's: superpoint_extractor.py:299-314,
// net/backbone.py:37-103); the program prints keypoint / match counts and a checksum and exits non-zero on any ABI error.
// With IN.bin / OUT.bin every weight tensor and input is read from IN.bin instead (raw little-endian fp32 / uint8 arrays in
// exactly the order this program would otherwise draw them: per network, per layer w, b, [gamma, beta, mean, var]; then events,
// mask, image) and the results are written to OUT.bin (per side counts, positions, sparse descriptors; then matches0 and the
// match counts): tests/test_boundary_gpu.py writes IN.bin from a seeded state dict and compares OUT.bin bit for bit with the oracle.
// With a fourth argument `lg` the same features also go through a 3-layer LightGlue (einx_lightglue: lightglue.py:522-716; the
// structs of einx.h filled in from C): its weights follow the inputs in IN.bin (per layer the fields of einx_lg_layer in their
// order, each matrix then its bias; then posenc.Wr, final_proj, matchability) and OUT.bin ends with its matches0 / matches1 / scores0.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "einx.h"

#define HIPCHK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)
#define EINXCHK(x)                                                                  \
  do {                                                                              \
    int s_ = (x);                                                                   \
    if (s_ != EINX_OK) {                                                            \
      fprintf(stderr, "%s:%d: einx status %d: %s\n", __FILE__, __LINE__, s_, einx_last_error()); \
      exit(3);                                                                      \
    }                                                                               \
  } while (0)

static FILE* g_in = nullptr;  // IN.bin, when given
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static float urand() {  // splitmix64 -> [0,1)
  uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// the next array of the program's fixed drawing order: from IN.bin when given (returns true), else left to the caller's generator
template <typename T>
static bool from_file(std::vector<T>& v) {
  if (!g_in) return false;
  if (fread(v.data(), sizeof(T), v.size(), g_in) != v.size()) {
    fprintf(stderr, "IN.bin ends early (array of %zu elements)\n", v.size());
    exit(2);
  }
  return true;
}

template <typename T>
static T* dalloc(size_t n) {
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, n * sizeof(T) + 256));
  return (T*)p;
}

static float* upload(const std::vector<float>& h) {
  float* d = dalloc<float>(h.size());
  HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

struct Layer {
  einx_conv_desc d;
};

// one conv block: seeded OIHW weights -> einx_conv_repack; optional BatchNorm(eval) -> einx_bn_fold
static Layer make_layer(int cin, int cout, int ks, bool relu, bool bn, bool pool, hipStream_t st) {
  std::vector<float> w((size_t)cout * cin * ks * ks), b(cout);
  const float a = sqrtf(6.0f / (float)(cin * ks * ks));
  if (!from_file(w))
    for (auto& v : w) v = (2.0f * urand() - 1.0f) * a;
  if (!from_file(b))
    for (auto& v : b) v = (2.0f * urand() - 1.0f) * 0.1f;
  float* w_oihw = upload(w);
  float* w_native = dalloc<float>(einx_conv_weight_elems(cin, cout, ks));
  EINXCHK(einx_conv_repack(w_oihw, cin, cout, ks, w_native, st));
  Layer L{};
  L.d.w_native = w_native;
  L.d.bias = upload(b);
  L.d.cin = cin;
  L.d.cout = cout;
  L.d.ks = ks;
  L.d.relu = relu;
  L.d.pool = pool;
  if (bn) {
    std::vector<float> g(cout), be(cout), mu(cout), var(cout);
    if (g_in) {
      from_file(g), from_file(be), from_file(mu), from_file(var);
    } else {
      for (int i = 0; i < cout; ++i) {
        g[i] = 0.5f + urand();
        be[i] = (2.0f * urand() - 1.0f) * 0.1f;
        mu[i] = (2.0f * urand() - 1.0f) * 0.3f;
        var[i] = 0.5f + urand();
      }
    }
    float* sc = dalloc<float>(cout);
    float* sh = dalloc<float>(cout);
    EINXCHK(einx_bn_fold(upload(g), upload(be), upload(mu), upload(var), 1e-5f, cout, sc, sh, st));
    L.d.scale = sc;
    L.d.shift = sh;
  }
  return L;
}

struct Net {
  std::vector<einx_conv_desc> bb, det, desc;
  einx_extractor* h = nullptr;
};

static Net make_net(int cin, bool bn, bool dilate, float input_div, hipStream_t st) {
  Net n;
  const int ch[5] = {cin, 64, 64, 128, 128};
  for (int s = 0; s < 4; ++s) {
    n.bb.push_back(make_layer(ch[s], ch[s + 1], 3, true, bn, false, st).d);
    n.bb.push_back(make_layer(ch[s + 1], ch[s + 1], 3, true, bn, s < 3, st).d);  // pool after stages 1-3
  }
  n.det.push_back(make_layer(128, 256, 3, true, bn, false, st).d);
  n.det.push_back(make_layer(256, 65, 1, false, bn, false, st).d);
  n.desc.push_back(make_layer(128, 256, 3, true, bn, false, st).d);
  n.desc.push_back(make_layer(256, 256, 1, false, bn, false, st).d);
  einx_extractor_desc d{};
  d.struct_size = sizeof d;  // checked by the library: a header / library mismatch is an error, not garbage
  d.cell = 8;
  d.n_backbone = (int)n.bb.size();
  d.n_det = (int)n.det.size();
  d.n_desc = (int)n.desc.size();
  d.backbone = n.bb.data();
  d.det_head = n.det.data();
  d.desc_head = n.desc.data();
  d.dilate_mask = dilate;
  d.border = 4;
  d.nms_radius = 4;
  d.top_k = 1024;
  d.det_thr = 1.0f;
  d.ordering_xy = 0;
  d.desc_scale = 1.0f;
  d.input_div = input_div;
  n.h = einx_extractor_create(&d);
  if (!n.h) {
    fprintf(stderr, "einx_extractor_create: %s\n", einx_last_error());
    exit(3);
  }
  return n;
}

// ---- LightGlue: the weight structs of einx.h filled in by a C program ---------------------------------------------------
static const float* lg_array(size_t n, float amp) {
  std::vector<float> h(n);
  if (!from_file(h))
    for (auto& v : h) v = (2.0f * urand() - 1.0f) * amp;
  return upload(h);
}

struct LgModel {
  std::vector<einx_lg_layer> layers;
  einx_lg_weights w{};
};

static void make_lightglue(LgModel& m, int n_layers, int heads, int d, int input_dim) {
  const float a1 = sqrtf(3.0f / (float)d), a2 = sqrtf(3.0f / (float)(2 * d));
  auto lin = [&](const float*& w, const float*& b, int out, int in, float amp) {  // nn.Linear: weight [out,in], bias [out]
    w = lg_array((size_t)out * in, amp);
    b = lg_array((size_t)out, 0.05f);
  };
  auto ffn = [&](const float*& w0, const float*& b0, const float*& g, const float*& be, const float*& w3, const float*& b3) {
    lin(w0, b0, 2 * d, 2 * d, a2);        // ffn.0
    g = lg_array((size_t)2 * d, 1.0f);   // ffn.1 (LayerNorm) weight, bias
    be = lg_array((size_t)2 * d, 0.05f);
    lin(w3, b3, d, 2 * d, a2);            // ffn.3
  };
  m.layers.resize(n_layers);
  for (auto& L : m.layers) {
    lin(L.Wqkv, L.bqkv, 3 * d, d, a1);  // SelfBlock (lightglue.py:240-272)
    lin(L.Wo, L.bo, d, d, a1);
    ffn(L.sf0_w, L.sf0_b, L.sln_g, L.sln_b, L.sf3_w, L.sf3_b);
    lin(L.Wqk, L.bqk, d, d, a1);        // CrossBlock (:275-330)
    lin(L.Wv, L.bv, d, d, a1);
    lin(L.Wco, L.bco, d, d, a1);
    ffn(L.cf0_w, L.cf0_b, L.cln_g, L.cln_b, L.cf3_w, L.cf3_b);
  }
  m.w.struct_size = sizeof(einx_lg_weights);
  m.w.layer_size = sizeof(einx_lg_layer);
  m.w.in_w = m.w.in_b = nullptr;  // input_dim == d: Identity
  m.w.Wr = lg_array((size_t)(d / heads / 2) * 2, 1.0f);  // posenc.Wr [head_dim/2, 2]
  lin(m.w.proj_w, m.w.proj_b, d, d, a1);                 // log_assignment[last].final_proj
  lin(m.w.match_w, m.w.match_b, 1, d, a1);               // ... .matchability
  m.w.n_layers = n_layers;
  m.w.heads = heads;
  m.w.d = d;
  m.w.input_dim = input_dim;
  m.w.filter_threshold = 0.0f;
  m.w.layers = m.layers.data();
}

struct Outs {
  einx_extract_out o{};
  einx_extract_shapes_t sh{};
  void* ws = nullptr;
  size_t ws_bytes = 0;
};

static Outs make_outs(const Net& n, int B, int H, int W) {
  Outs r;
  EINXCHK(einx_extract_shapes(n.h, H, W, &r.sh));
  const auto& s = r.sh;
  const size_t hw = (size_t)s.hc * s.wc;
  r.o.feats = dalloc<float>((size_t)B * s.feat_channels * hw);
  r.o.logits = dalloc<float>((size_t)B * s.det_channels * hw);
  r.o.raw = dalloc<float>((size_t)B * s.desc_dim * hw);
  r.o.prob = dalloc<float>((size_t)B * s.det_channels * hw);
  r.o.score = dalloc<float>((size_t)B * s.Hp * s.Wp);
  r.o.coarse = dalloc<float>((size_t)B * s.desc_dim * hw);
  r.o.raw_cl = dalloc<float>((size_t)B * s.desc_dim * hw);
  r.o.nms = dalloc<float>((size_t)B * H * W);
  r.o.positions = dalloc<float>((size_t)B * s.cap * 3);
  r.o.indices = dalloc<int32_t>((size_t)B * s.cap);
  r.o.counts = dalloc<int32_t>(B);
  r.o.thr = dalloc<float>(B);
  r.o.not_converged = dalloc<int32_t>(B);
  r.o.sparse_desc = dalloc<float>((size_t)B * s.cap * s.desc_dim);
  r.o.cap = s.cap;
  r.ws_bytes = einx_extract_ws_bytes(n.h, B, H, W, s.cap, 8);
  HIPCHK(hipMalloc(&r.ws, r.ws_bytes));
  return r;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 2;
  const int H = 260, W = 346, CE = 5;
  const char* out_path = argc > 3 ? argv[3] : nullptr;
  const bool with_lg = argc > 4 && argv[4][0] == 'l';
  if (argc > 3 && !(g_in = fopen(argv[2], "rb"))) {
    fprintf(stderr, "cannot open %s\n", argv[2]);
    return 2;
  }
  printf("%s, %d HIP device(s)\n", einx_version(), einx_device_count());
  if (einx_abi_version() != EINX_ABI_VERSION) {
    fprintf(stderr, "libeinx_hip.so speaks ABI %d, this host was built against %d\n", einx_abi_version(), EINX_ABI_VERSION);
    return 1;
  }
  if (einx_device_count() < 1) {
    fprintf(stderr, "no HIP device: the library has no CPU path\n");
    return 1;
  }
  hipStream_t st;
  HIPCHK(hipStreamCreate(&st));
  Net ev_net = make_net(CE, /*bn*/ true, /*dilate*/ true, 0.0f, st);   // VGGExtractor (event voxel grid)
  Net im_net = make_net(1, /*bn*/ false, /*dilate*/ false, 255.0f, st);  // SuperPointv1 (gray image, scaled in place)
  // synthetic inputs: sparse event voxels with their support mask, smooth-ish gray image
  std::vector<float> ev((size_t)B * CE * H * W, 0.0f), img((size_t)B * H * W);
  std::vector<uint8_t> mask((size_t)B * H * W, 0);
  if (g_in) {
    from_file(ev), from_file(mask), from_file(img);
  } else {
    for (int b = 0; b < B; ++b)
      for (int p = 0; p < H * W; ++p) {
        if (urand() < 0.1f) {
          mask[(size_t)b * H * W + p] = 1;
          for (int c = 0; c < CE; ++c) ev[((size_t)b * CE + c) * H * W + p] = 2.0f * urand() - 1.0f;
        }
        img[(size_t)b * H * W + p] = floorf(urand() * 256.0f);
      }
  }
  float* d_ev = upload(ev);
  float* d_img = upload(img);
  uint8_t* d_mask = dalloc<uint8_t>(mask.size());
  HIPCHK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
  Outs eo = make_outs(ev_net, B, H, W), io = make_outs(im_net, B, H, W);
  // ---- the hot path: two einx_extract calls + the matcher, all enqueued on one stream, no host synchronisation in between
  EINXCHK(einx_extract(ev_net.h, d_ev, d_mask, B, H, W, 8, eo.ws, eo.ws_bytes, &eo.o, st));
  EINXCHK(einx_extract(im_net.h, d_img, nullptr, B, H, W, 8, io.ws, io.ws_bytes, &io.o, st));
  const int cap0 = eo.sh.cap, cap1 = io.sh.cap, D = eo.sh.desc_dim;
  void* mws = nullptr;
  HIPCHK(hipMalloc(&mws, einx_mnn_ws_bytes(B, cap0, cap1)));
  int64_t* m0 = dalloc<int64_t>((size_t)B * cap0);
  int64_t* m1 = dalloc<int64_t>((size_t)B * cap1);
  float* s0 = dalloc<float>((size_t)B * cap0);
  float* s1 = dalloc<float>((size_t)B * cap1);
  EINXCHK(einx_mnn(eo.o.sparse_desc, eo.o.counts, cap0, io.o.sparse_desc, io.o.counts, cap1, B, D, mws, m0, m1, s0, s1, nullptr, st));
  float* mk0 = dalloc<float>((size_t)B * cap0 * 3);
  float* mk1 = dalloc<float>((size_t)B * cap0 * 3);
  int32_t* nmatch = dalloc<int32_t>(B);
  EINXCHK(einx_gather_matches(eo.o.positions, io.o.positions, m0, eo.o.counts, cap0, cap1, B, 3, mk0, mk1, nmatch, st));
  // ---- optionally the learned matcher on the same features (device-side counts, no host synchronisation either)
  int64_t *lm0 = nullptr, *lm1 = nullptr;
  float *ls0 = nullptr, *ls1 = nullptr;
  LgModel lgm;
  if (with_lg) {
    make_lightglue(lgm, /*n_layers*/ 3, /*heads*/ 4, D, D);
    const size_t lws = einx_lg_ws_bytes_heads(B, cap0, cap1, D, 4, D);
    if (!lws) {
      fprintf(stderr, "einx_lg_ws_bytes_heads: unsupported widths\n");
      return 3;
    }
    void* lgws = nullptr;
    HIPCHK(hipMalloc(&lgws, lws));
    lm0 = dalloc<int64_t>((size_t)B * cap0);
    lm1 = dalloc<int64_t>((size_t)B * cap1);
    ls0 = dalloc<float>((size_t)B * cap0);
    ls1 = dalloc<float>((size_t)B * cap1);
    EINXCHK(einx_lightglue(&lgm.w, eo.o.positions, eo.o.sparse_desc, eo.o.counts, cap0, io.o.positions, io.o.sparse_desc, io.o.counts, cap1, B,
                           (float)H, (float)W, (float)H, (float)W, lgws, lm0, lm1, ls0, ls1, /*la*/ nullptr, /*ref*/ nullptr, nullptr, 0, st));
  }
  HIPCHK(hipStreamSynchronize(st));
  std::vector<int32_t> n0(B), n1(B), nm(B), bad(2 * B);
  HIPCHK(hipMemcpy(n0.data(), eo.o.counts, B * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n1.data(), io.o.counts, B * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(nm.data(), nmatch, B * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(bad.data(), eo.o.not_converged, B * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(bad.data() + B, io.o.not_converged, B * 4, hipMemcpyDeviceToHost));
  std::vector<float> pos((size_t)B * cap0 * 3), desc((size_t)cap0 * D);
  HIPCHK(hipMemcpy(pos.data(), eo.o.positions, pos.size() * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(desc.data(), eo.o.sparse_desc, desc.size() * 4, hipMemcpyDeviceToHost));
  int rc = 0;
  for (int b = 0; b < B; ++b) {
    printf("pair %d: %d event keypoints, %d image keypoints, %d mutual matches (cap %d)\n", b, n0[b], n1[b], nm[b], cap0);
    if (n0[b] <= 0 || n0[b] > cap0 || n1[b] <= 0 || n1[b] > cap1 || nm[b] < 0 || nm[b] > n0[b] || bad[b] || bad[B + b]) rc = 4;
    // raster order, inside the image, away from the 4-pixel border
    for (int i = 0; i < n0[b]; ++i) {
      const float y = pos[((size_t)b * cap0 + i) * 3], x = pos[((size_t)b * cap0 + i) * 3 + 1];
      if (y < 2.0f || y > H - 2.0f || x < 1.0f || x > W - 1.0f) rc = 5;
      if (i > 0) {
        const float py = pos[((size_t)b * cap0 + i - 1) * 3], px = pos[((size_t)b * cap0 + i - 1) * 3 + 1];
        if (!(y > py || (y == py && x > px))) rc = 6;
      }
    }
  }
  if (out_path) {  // everything a caller of the reference reads from the sparse outputs, as raw arrays
    FILE* fo = fopen(out_path, "wb");
    if (!fo) {
      fprintf(stderr, "cannot write %s\n", out_path);
      return 2;
    }
    auto dump = [&](const void* dptr, size_t bytes) {
      std::vector<char> h(bytes);
      HIPCHK(hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost));
      fwrite(h.data(), 1, bytes, fo);
    };
    const Outs* sides[2] = {&eo, &io};
    for (const Outs* o : sides) {
      dump(o->o.counts, (size_t)B * 4);
      dump(o->o.positions, (size_t)B * o->sh.cap * 3 * 4);
      dump(o->o.sparse_desc, (size_t)B * o->sh.cap * o->sh.desc_dim * 4);
    }
    dump(m0, (size_t)B * cap0 * 8);
    dump(nmatch, (size_t)B * 4);
    if (with_lg) {
      dump(lm0, (size_t)B * cap0 * 8);
      dump(lm1, (size_t)B * cap1 * 8);
      dump(ls0, (size_t)B * cap0 * 4);
    }
    fclose(fo);
  }
  if (with_lg) {
    std::vector<int64_t> hm((size_t)B * cap0);
    HIPCHK(hipMemcpy(hm.data(), lm0, hm.size() * 8, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
      int cnt = 0;
      for (int i = 0; i < n0[b]; ++i) {
        const int64_t j = hm[(size_t)b * cap0 + i];
        if (j >= n1[b] || j < -1) rc = 8;  // a match index must point at a keypoint of the other side
        cnt += j >= 0;
      }
      printf("pair %d: %d LightGlue matches\n", b, cnt);
    }
  }
  double nrm = 0.0;
  for (int c = 0; c < D; ++c) nrm += (double)desc[c] * desc[c];
  printf("first event descriptor: squared norm %.6f (unit length expected)\n", nrm);
  if (nrm < 0.9999 || nrm > 1.0001) rc = 7;
  einx_extractor_destroy(ev_net.h);
  einx_extractor_destroy(im_net.h);
  printf(rc == 0 ? "C ABI host: OK\n" : "C ABI host: FAILED (%d)\n", rc);
  return rc;
}
