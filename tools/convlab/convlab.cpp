// convlab -- stand-alone timing / A-B harness for deva_conv2d (no Python, no torch: starts in milliseconds, so a
// rocprofv3 --pmc pass over it costs seconds).
//
//   convlab [--libs a.so,b.so,...] [--only substr] [--iters N] [--set frame480|big|small] [--check]
//
// Every library is dlopen'ed and driven through the C ABI of include/deva_hip.h.  The layer list is the 480p /
// 5-object frame of bench.py (profiles/r03c/conv_layers_480p5.json) with the number of calls per frame, so
// "frame" = sum(calls x time) is the convolution time of one propagated frame.  --check compares every
// library's output with the first library's (max abs / max rel difference).  Weights are packed by
// deva_conv_pack when the library exports it (layouts differ between kernel generations), otherwise [K][cout_pad].
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "deva_hip.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

// --keepalive: one wave that sleeps for `ticks` of the 100 MHz wall clock on a second stream, so that the GPU is never
// idle between the timed launches (does the clock / power management react to the gaps between kernels?)
__global__ void keepalive_kernel(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink) *sink = 1;
}
// --heat N: N workgroups of dependent FMAs beside the timed launches (power / clock probe)

struct Layer {
  const char* name;
  int c0, c1, cout, k, stride, batch, oh, ow;
  int relu_in, res, act;
  double calls;  // per frame (480p, 5 objects, memory frame every 5th)
  int set;       // 1 = big batch-5 layers, 2 = small / batch-1 layers
};

// 480p, 5 objects: (cin, cout, k, stride, batch, oh, ow, calls) from profiles/r03c/conv_layers_480p5.json; the
// flags (virtual concat, relu-on-load, residual, activation) follow deva/model/_graph.py.
static const Layer kLayers[] = {
    {"up8_4 3x3 256>256 @120x216 x5", 256, 0, 256, 3, 1, 5, 120, 216, 1, 1, 0, 2.0, 1},
    {"gru 3x3 512+512>1536 @30x54 x5", 512, 512, 1536, 3, 1, 5, 30, 54, 0, 0, 0, 1.2, 1},
    {"fuser 3x3 512>512 @30x54 x5", 512, 0, 512, 3, 1, 5, 30, 54, 1, 1, 0, 4.6, 1},
    {"up16_8 3x3 512>256 @60x108 x5", 512, 0, 256, 3, 1, 5, 60, 108, 1, 0, 0, 1.0, 1},
    {"up16_8 3x3 256>256 @60x108 x5", 256, 0, 256, 3, 1, 5, 60, 108, 1, 1, 0, 1.0, 1},
    {"res3 3x3 256>256 @30x54 x1", 256, 0, 256, 3, 1, 1, 30, 54, 0, 0, 1, 5.0, 2},
    {"fuserx 3x3 512>512 @30x54 x1", 512, 0, 512, 3, 1, 1, 30, 54, 1, 0, 0, 1.2, 2},
    {"res3 1x1 256>1024 @30x54 x1", 256, 0, 1024, 1, 1, 1, 30, 54, 0, 1, 1, 6.0, 2},
    {"res3 1x1 1024>256 @30x54 x1", 1024, 0, 256, 1, 1, 1, 30, 54, 0, 0, 1, 5.0, 2},
    {"res2 3x3 128>128 @60x108 x1", 128, 0, 128, 3, 1, 1, 60, 108, 0, 0, 1, 3.0, 2},
    {"res1 3x3 64>64 @120x216 x1", 64, 0, 64, 3, 1, 1, 120, 216, 0, 0, 1, 3.0, 2},
    {"res1 1x1 64>256 @120x216 x1", 64, 0, 256, 1, 1, 1, 120, 216, 0, 1, 1, 4.0, 2},
    {"fuserds 1x1 512>512 @30x54 x5", 512, 0, 512, 1, 1, 5, 30, 54, 0, 1, 0, 2.0, 2},
    {"stem 7x7s2 3>64 @240x432 x1", 3, 0, 64, 7, 2, 1, 240, 432, 0, 0, 1, 1.2, 2},
    {"up16_8ds 1x1 512>256 @60x108 x5", 512, 0, 256, 1, 1, 5, 60, 108, 0, 0, 0, 1.0, 2},
    {"res2 1x1 128>512 @60x108 x1", 128, 0, 512, 1, 1, 1, 60, 108, 0, 1, 1, 4.0, 2},
    {"menc 3x3 64>64 @120x216 x5", 64, 0, 64, 3, 1, 5, 120, 216, 0, 1, 1, 0.8, 2},
    {"res2 1x1 512>128 @60x108 x1", 512, 0, 128, 1, 1, 1, 60, 108, 0, 0, 1, 3.0, 2},
    {"pred 3x3 256>1 @120x216 x5", 256, 0, 1, 3, 1, 5, 120, 216, 1, 0, 0, 1.0, 2},
    {"menc 3x3 128>128 @60x108 x5", 128, 0, 128, 3, 1, 5, 60, 108, 0, 1, 1, 0.6, 2},
    {"cbam 7x7 2>1 @30x54 x5", 2, 0, 1, 7, 1, 5, 30, 54, 0, 0, 0, 1.2, 2},
    {"proj12 1x1 1024>1024 @30x54 x1", 1024, 0, 1024, 1, 1, 1, 30, 54, 0, 0, 0, 1.0, 2},
    {"menc 3x3 256>256 @30x54 x5", 256, 0, 256, 3, 1, 5, 30, 54, 0, 1, 1, 0.6, 2},
    {"key 3x3 512>64 @30x54 x1", 512, 0, 64, 3, 1, 1, 30, 54, 0, 0, 0, 2.0, 2},
    {"scomp 1x1 512+1>512 @30x54 x5", 512, 1, 512, 1, 1, 5, 30, 54, 0, 1, 0, 1.0, 2},
    {"dfp0 1x1 512>512 @60x108 x1", 512, 0, 512, 1, 1, 1, 60, 108, 0, 0, 0, 1.0, 2},
    {"dfp1 1x1 256>256 @120x216 x1", 256, 0, 256, 1, 1, 1, 120, 216, 0, 0, 0, 1.0, 2},
    {"res2 3x3s2 128>128 @60x108 x1", 128, 0, 128, 3, 2, 1, 60, 108, 0, 0, 1, 1.0, 2},
    {"shrink 3x3 512>1 @30x54 x1", 512, 0, 1, 3, 1, 1, 30, 54, 0, 0, 3, 1.0, 2},
    {"g8 1x1 256>512 @30x54 x5", 256, 0, 512, 1, 1, 5, 30, 54, 0, 1, 0, 1.2, 2},
    {"res3 3x3s2 256>256 @30x54 x1", 256, 0, 256, 3, 2, 1, 30, 54, 0, 0, 1, 1.0, 2},
    {"res2ds 1x1s2 256>512 @60x108 x1", 256, 0, 512, 1, 2, 1, 60, 108, 0, 0, 0, 1.0, 2},
    {"res1 1x1 256>64 @120x216 x1", 256, 0, 64, 1, 1, 1, 120, 216, 0, 0, 1, 2.0, 2},
    {"fuser2 3x3 256>512 @30x54 x5", 256, 0, 512, 3, 1, 5, 30, 54, 1, 0, 0, 0.2, 2},
    {"fuserxds 1x1 512>512 @30x54 x1", 512, 0, 512, 1, 1, 1, 30, 54, 0, 0, 0, 1.2, 2},
    {"res3ds 1x1s2 512>1024 @30x54 x1", 512, 0, 1024, 1, 2, 1, 30, 54, 0, 0, 0, 1.0, 2},
    {"g4 1x1 256+1>512 @30x54 x5", 256, 1, 512, 1, 1, 5, 30, 54, 0, 1, 0, 1.0, 2},
    {"res2 1x1 256>128 @120x216 x1", 256, 0, 128, 1, 1, 1, 120, 216, 0, 0, 1, 1.0, 2},
    {"res3 1x1 512>256 @60x108 x1", 512, 0, 256, 1, 1, 1, 60, 108, 0, 0, 1, 1.0, 2},
    {"res1 1x1 64>64 @120x216 x1", 64, 0, 64, 1, 1, 1, 120, 216, 0, 0, 1, 1.0, 2},
    {"mstem 7x7s2 1>64 @240x432 x5", 1, 0, 64, 7, 2, 5, 240, 432, 0, 1, 0, 0.2, 2},
    {"menc 3x3s2 64>128 @60x108 x5", 64, 0, 128, 3, 2, 5, 60, 108, 0, 0, 1, 0.2, 2},
    {"menc 3x3s2 128>256 @30x54 x5", 128, 0, 256, 3, 2, 5, 30, 54, 0, 0, 1, 0.2, 2},
    {"mencds 1x1s2 64>128 @60x108 x5", 64, 0, 128, 1, 2, 5, 60, 108, 0, 0, 0, 0.2, 2},
    {"mencds 1x1s2 128>256 @30x54 x5", 128, 0, 256, 1, 2, 5, 30, 54, 0, 0, 0, 0.2, 2},
};

typedef int (*conv_fn)(const deva_conv_desc*, void*);
typedef int64_t (*pack_fn)(const float*, float*, int, int, int, int, int, int*, int*);
typedef const char* (*err_fn)(void);
typedef int64_t (*pack16_fn)(const float*, uint16_t*, int, int, int, int, int*);
typedef int64_t (*packsp_fn)(const float*, uint16_t*, int, int, int, int, int*, int*);
typedef int64_t (*packwi_fn)(const float*, float*, int, int);

struct Lib {
  std::string path;
  void* h;
  conv_fn conv;
  pack_fn pack;  // optional (newer libraries): deva_conv_pack of include/deva_hip.h
  err_fn err;
  pack16_fn pack16;  // optional: deva_conv_pack_f16 (amp path)
  packsp_fn packsp;  // optional: deva_conv_pack_split (fp32-accurate hi/lo split on the f16 pipes)
  packwi_fn packwi;  // optional: deva_conv_pack_wino (fp32 Winograd F(2x2, 3x3))
};

static constexpr int64_t kGuard = 8192;  // like deva/hip/ops.py:_alloc

static float* dev_alloc_guarded(int64_t n, std::vector<void*>& keep) {
  float* p;
  HIP_OK(hipMalloc(&p, (n + 2 * kGuard) * sizeof(float)));
  HIP_OK(hipMemset(p, 0, (n + 2 * kGuard) * sizeof(float)));
  keep.push_back(p);
  return p + kGuard;
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) {
    s = s * 1664525u + 1013904223u;
    x = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
  }
}

int main(int argc, char** argv) {
  std::string libs = "tracking-anything-with-deva_amd/deva/hip/libdeva_hip.so";
  std::string only, set = "frame480", shapes;
  int iters = 20;
  bool check = false, csv = false, stamps = false, amp = false, split = false, overflow = false, zero_in = false, wino = false;
  int keepalive_ms = 0;
  int rounds = 1;
  double warm_ms = 15.0, time_ms = 20.0;
  int q4 = 1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a == "--libs" && i + 1 < argc) libs = argv[++i];
    else if (a == "--only" && i + 1 < argc) only = argv[++i];
    else if (a == "--iters" && i + 1 < argc) iters = atoi(argv[++i]);
    else if (a == "--set" && i + 1 < argc) set = argv[++i];
    else if (a == "--check") check = true;
    else if (a == "--noq4") q4 = 0;
    else if (a == "--csv") csv = true;
    else if (a == "--stamps") stamps = true;
    else if (a == "--amp") amp = true;  // fp16 operands on every library but the first (which stays the fp32 reference)
    else if (a == "--wino") wino = true;  // fp32 Winograd on every library but the first
    else if (a == "--split") split = true;  // hi/lo fp16 split (fp32-accurate) on every library but the first
    else if (a == "--zero_in") zero_in = true;  // all-zero activations (power probe: operand switching activity)
    else if (a == "--overflow") overflow = true;  // one input element beyond the fp16 range: the split path must fall back
    else if (a == "--keepalive" && i + 1 < argc) keepalive_ms = atoi(argv[++i]);
    else if (a == "--warm_ms" && i + 1 < argc) warm_ms = atof(argv[++i]);
    else if (a == "--time_ms" && i + 1 < argc) time_ms = atof(argv[++i]);
    else if (a == "--shapes" && i + 1 < argc) shapes = argv[++i];
    else if (a == "--rounds" && i + 1 < argc) rounds = atoi(argv[++i]);  // A,B,A,B,... timing rounds per layer, median reported
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  std::vector<Layer> layers(std::begin(kLayers), std::end(kLayers));
  static std::vector<std::string> custom_names;
  if (!shapes.empty()) {  // --shapes "c0,c1,cout,k,stride,batch,oh,ow,relu,res,act[,calls];..."
    layers.clear();
    custom_names.reserve(64);
    for (size_t p = 0; p < shapes.size();) {
      size_t q = shapes.find(';', p);
      if (q == std::string::npos) q = shapes.size();
      const std::string one = shapes.substr(p, q - p);
      int v[11] = {0};
      float calls = 1.0f;  // optional 12th field: calls per frame
      if (sscanf(one.c_str(), "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%f", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8, v + 9,
                 v + 10, &calls) >= 8) {
        custom_names.push_back(one);
        std::replace(custom_names.back().begin(), custom_names.back().end(), ',', '_');  // (the CSV output is comma-separated)
        layers.push_back(Layer{custom_names.back().c_str(), v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], calls, 1});
      }
      p = q + 1;
    }
  }
  std::vector<Lib> L;
  for (size_t p = 0; p < libs.size();) {
    size_t q = libs.find(',', p);
    if (q == std::string::npos) q = libs.size();
    Lib l;
    l.path = libs.substr(p, q - p);
    l.h = dlopen(l.path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!l.h) { fprintf(stderr, "dlopen %s: %s\n", l.path.c_str(), dlerror()); return 1; }
    l.conv = (conv_fn)dlsym(l.h, "deva_conv2d");
    l.pack = (pack_fn)dlsym(l.h, "deva_conv_pack");
    l.err = (err_fn)dlsym(l.h, "deva_hip_last_error");
    l.pack16 = (pack16_fn)dlsym(l.h, "deva_conv_pack_f16");
    l.packsp = (packsp_fn)dlsym(l.h, "deva_conv_pack_split");
    l.packwi = (packwi_fn)dlsym(l.h, "deva_conv_pack_wino");
    if (!l.conv) { fprintf(stderr, "%s: no deva_conv2d\n", l.path.c_str()); return 1; }
    L.push_back(l);
    p = q + 1;
  }
  hipStream_t st, st2;
  HIP_OK(hipStreamCreate(&st));
  HIP_OK(hipStreamCreate(&st2));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  float* ws;
  const int64_t ws_elems = 16ll << 20;
  HIP_OK(hipMalloc(&ws, ws_elems * sizeof(float)));

  printf("%-36s", "layer");
  for (auto& l : L) {
    std::string b = l.path.substr(l.path.find_last_of('/') + 1);
    printf(" | %22s", b.c_str());
  }
  printf("\n");
  std::vector<double> frame_us(L.size(), 0.0), big_us(L.size(), 0.0), small_us(L.size(), 0.0);
  double frame_gf = 0.0;
  for (const Layer& ly : layers) {
    if (!only.empty() && !strstr(ly.name, only.c_str())) continue;
    if (set == "big" && ly.set != 1) continue;
    if (set == "small" && ly.set != 2) continue;
    std::vector<void*> keep;
    const int cin = ly.c0 + ly.c1, H = ly.oh * ly.stride, W = ly.ow * ly.stride, pad = ly.k / 2;
    const int64_t in0_n = (int64_t)ly.batch * ly.c0 * H * W, in1_n = (int64_t)ly.batch * ly.c1 * H * W;
    const int64_t out_n = (int64_t)ly.batch * ly.cout * ly.oh * ly.ow;
    std::vector<float> h_in0(in0_n), h_in1(in1_n), h_w((int64_t)ly.cout * cin * ly.k * ly.k), h_b(ly.cout), h_res(out_n);
    fill(h_in0, 1, 1.0f);
    fill(h_in1, 2, 1.0f);
    fill(h_w, 3, sqrtf(6.0f / (cin * ly.k * ly.k)));
    fill(h_b, 4, 0.1f);
    fill(h_res, 5, 1.0f);
    if (zero_in) {
      std::fill(h_in0.begin(), h_in0.end(), 0.0f);
      std::fill(h_in1.begin(), h_in1.end(), 0.0f);
    }
    if (overflow) h_in0[h_in0.size() / 3] = 1.0e6f;  // beyond fp16: the split path raises its flag, the gated fp32 kernels redo the layer
    float* d_in0 = dev_alloc_guarded(in0_n, keep);
    float* d_in1 = ly.c1 ? dev_alloc_guarded(in1_n, keep) : nullptr;
    float* d_res = ly.res ? dev_alloc_guarded(out_n, keep) : nullptr;
    float* d_b = dev_alloc_guarded(ly.cout, keep);
    HIP_OK(hipMemcpy(d_in0, h_in0.data(), in0_n * 4, hipMemcpyHostToDevice));
    if (d_in1) HIP_OK(hipMemcpy(d_in1, h_in1.data(), in1_n * 4, hipMemcpyHostToDevice));
    if (d_res) HIP_OK(hipMemcpy(d_res, h_res.data(), out_n * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_b, h_b.data(), ly.cout * 4, hipMemcpyHostToDevice));
    const double gf = 2.0 * cin * ly.k * ly.k * ly.cout * (double)ly.batch * ly.oh * ly.ow / 1e9;
    frame_gf += gf * ly.calls;
    printf("%-36s", ly.name);
    std::vector<float> ref;
    std::vector<deva_conv_desc> descs(L.size());
    std::vector<int> n_time_of(L.size(), 0), n_warm_of(L.size(), 0);
    std::vector<std::vector<double>> samples(L.size());
    std::vector<std::string> tails(L.size());
    for (size_t li = 0; li < L.size(); ++li) {
      Lib& l = L[li];
      // weights in this library's layout
      const int cout_pad_default = (ly.cout + 31) / 32 * 32;
      const int K = ly.k * ly.k * cin;
      int k_layout = DEVA_KLAYOUT_TAP_MAJOR, cout_pad = cout_pad_default;
      std::vector<float> packed;
      if (l.pack) {
        const int64_t n = l.pack(h_w.data(), nullptr, ly.cout, cin, ly.k, ly.k, q4, &k_layout, &cout_pad);
        if (n <= 0) { printf(" | pack failed"); continue; }
        packed.assign(n, 0.0f);
        l.pack(h_w.data(), packed.data(), ly.cout, cin, ly.k, ly.k, q4, &k_layout, &cout_pad);
      } else {
        packed.assign((int64_t)K * cout_pad, 0.0f);
        const bool chunk = ly.k > 1 && cin % 32 == 0;
        k_layout = chunk ? DEVA_KLAYOUT_CHUNK32 : DEVA_KLAYOUT_TAP_MAJOR;
        const int taps = ly.k * ly.k;
        for (int m = 0; m < ly.cout; ++m)
          for (int c = 0; c < cin; ++c)
            for (int t = 0; t < taps; ++t) {
              const int k = chunk ? ((c / 32) * taps + t) * 32 + c % 32 : t * cin + c;
              packed[(int64_t)k * cout_pad + m] = h_w[((int64_t)m * cin + c) * taps + t];
            }
      }
      float* d_w = dev_alloc_guarded(packed.size(), keep);
      HIP_OK(hipMemcpy(d_w, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
      float* d_out = dev_alloc_guarded(out_n, keep);
      deva_conv_desc d;
      memset(&d, 0, sizeof d);
      d.in0 = d_in0;
      d.in1 = d_in1;
      d.in0_batch_stride = (int64_t)ly.c0 * H * W;
      d.in1_batch_stride = (int64_t)ly.c1 * H * W;
      d.c0 = ly.c0;
      d.c1 = ly.c1;
      d.batch = ly.batch;
      d.height = H;
      d.width = W;
      d.weight = d_w;
      d.bias = d_b;
      d.cout = ly.cout;
      d.cout_pad = cout_pad;
      d.k_layout = k_layout;
      d.kh = d.kw = ly.k;
      d.stride = ly.stride;
      d.pad = pad;
      d.relu_in = ly.relu_in;
      d.residual = d_res;
      d.residual_batch_stride = (int64_t)ly.cout * ly.oh * ly.ow;
      d.act = ly.act;
      d.out = d_out;
      d.in_guard_elems = (int32_t)kGuard;
      d.workspace = ws;
      d.workspace_elems = ws_elems;
      if (amp && li > 0 && l.pack16) {
        int cp = 0;
        const int64_t n16 = l.pack16(h_w.data(), nullptr, ly.cout, cin, ly.k, ly.k, &cp);
        if (n16 > 0) {
          std::vector<uint16_t> w16(n16);
          l.pack16(h_w.data(), w16.data(), ly.cout, cin, ly.k, ly.k, &cp);
          float* d_w16 = dev_alloc_guarded((n16 + 1) / 2, keep);
          HIP_OK(hipMemcpy(d_w16, w16.data(), n16 * 2, hipMemcpyHostToDevice));
          d.weight_f16 = d_w16;
          d.amp = 1;
        }
      }
      if (split && li > 0 && l.packsp) {
        int cp = 0, e = 0;
        const int64_t n16 = l.packsp(h_w.data(), nullptr, ly.cout, cin, ly.k, ly.k, &cp, &e);
        if (n16 > 0) {
          std::vector<uint16_t> w16(n16);
          l.packsp(h_w.data(), w16.data(), ly.cout, cin, ly.k, ly.k, &cp, &e);
          float* d_w16 = dev_alloc_guarded((n16 + 1) / 2, keep);
          HIP_OK(hipMemcpy(d_w16, w16.data(), n16 * 2, hipMemcpyHostToDevice));
          d.weight_f16 = d_w16;
          d.amp = 2;
          d.split_scale_log2 = e;
          d.split_flag = (int32_t*)dev_alloc_guarded(4, keep);  // zeroed
        }
      }
      if (wino && li > 0 && l.packwi && ly.k == 3 && ly.stride == 1) {
        const int64_t nw = l.packwi(h_w.data(), nullptr, ly.cout, cin);
        if (nw > 0) {
          std::vector<float> ww(nw);
          l.packwi(h_w.data(), ww.data(), ly.cout, cin);
          float* d_ww = dev_alloc_guarded(nw, keep);
          HIP_OK(hipMemcpy(d_ww, ww.data(), nw * 4, hipMemcpyHostToDevice));
          d.weight_wino = d_ww;
        }
      }
      int rc = 0;
      for (int w = 0; w < 2 && !rc; ++w) rc = l.conv(&d, st);
      if (rc) { printf(" | error: %s", l.err ? l.err() : "?"); continue; }
      // calibrate, then warm up for warm_ms and time for >= time_ms WITHOUT an idle gap in between: after an idle period
      // the chip runs its first milliseconds at a lower clock (a 0.3 ms kernel measured right behind a synchronize is
      // 15-20 % slower than the same kernel in a busy stream)
      HIP_OK(hipEventRecord(e0, st));
      l.conv(&d, st);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      const double est_ms = std::max(1e-3, (double)ms);
      const int n_warm = (int)std::min(5000.0, std::max(1.0, warm_ms / est_ms));
      const int n_time = (int)std::min(20000.0, std::max((double)iters, time_ms / est_ms));
      if (keepalive_ms > 0) hipLaunchKernelGGL(keepalive_kernel, dim3(1), dim3(64), 0, st2, (long long)keepalive_ms * 100000ll, (int*)nullptr);
      for (int w = 0; w < n_warm; ++w) l.conv(&d, st);
      HIP_OK(hipEventRecord(e0, st));
      for (int it = 0; it < n_time; ++it) l.conv(&d, st);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      if (keepalive_ms > 0) HIP_OK(hipStreamSynchronize(st2));
      const double us = ms * 1e3 / n_time;
      samples[li].push_back(us);
      descs[li] = d;
      n_time_of[li] = n_time;
      n_warm_of[li] = n_warm;
      if (stamps) {  // probes library: wall-clock stamps of one more launch (entry, tile 0 staged, loop end, epilogue end)
        typedef int (*probe_fn)(unsigned long long*);
        probe_fn setp = (probe_fn)dlsym(l.h, "deva_conv_set_probe");
        if (setp) {
          const int nb = 65536;
          unsigned long long* dbuf;
          HIP_OK(hipMalloc(&dbuf, nb * 4 * 8));
          HIP_OK(hipMemset(dbuf, 0, nb * 4 * 8));
          setp(dbuf);
          l.conv(&d, st);
          HIP_OK(hipStreamSynchronize(st));
          setp(nullptr);
          std::vector<unsigned long long> h(nb * 4);
          HIP_OK(hipMemcpy(h.data(), dbuf, nb * 4 * 8, hipMemcpyDeviceToHost));
          HIP_OK(hipFree(dbuf));
          unsigned long long t0 = ~0ull, t3 = 0;
          int blocks = 0;
          double s01 = 0, s12 = 0, s23 = 0, first_end = 1e30, last_start = 0;
          for (int b = 0; b < nb; ++b)
            if (h[b * 4]) { ++blocks; t0 = std::min(t0, h[b * 4]); t3 = std::max(t3, h[b * 4 + 3]); }
          for (int b = 0; b < nb; ++b)
            if (h[b * 4]) {
              s01 += (double)(h[b * 4 + 1] - h[b * 4]); s12 += (double)(h[b * 4 + 2] - h[b * 4 + 1]); s23 += (double)(h[b * 4 + 3] - h[b * 4 + 2]);
              first_end = std::min(first_end, (double)(h[b * 4 + 3] - t0)); last_start = std::max(last_start, (double)(h[b * 4] - t0));
            }
          printf("\n    stamps: %d workgroups, kernel span %.2f us; mean prologue %.2f, loop %.2f, tail %.2f us; last start +%.2f, first end +%.2f",
                 blocks, (t3 - t0) * 0.01, s01 / blocks * 0.01, s12 / blocks * 0.01, s23 / blocks * 0.01, last_start * 0.01, first_end * 0.01);
        }
      }
      if (check) {
        std::vector<float> o(out_n);
        HIP_OK(hipMemcpy(o.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
        if (li == 0) {
          ref = o;
        } else {
          double ma = 0, mr = 0;
          for (int64_t i = 0; i < out_n; ++i) {
            const double dlt = fabs((double)o[i] - ref[i]);
            ma = std::max(ma, dlt);
            mr = std::max(mr, dlt / (fabs((double)ref[i]) + 1.0));
          }
          char buf[48];
          int flag = -1;
          if (descs[li].split_flag) HIP_OK(hipMemcpy(&flag, descs[li].split_flag, 4, hipMemcpyDeviceToHost));
          if (flag >= 0) snprintf(buf, sizeof buf, " d%.1e f%d", mr, flag);
          else snprintf(buf, sizeof buf, " d%.1e", mr);
          tails[li] = buf;
        }
      }
    }
    // further rounds: the libraries take turns (A, B, A, B, ...) so that slow drifts of the clock hit all of them alike
    for (int r = 1; r < rounds; ++r)
      for (size_t li = 0; li < L.size(); ++li) {
        if (samples[li].empty()) continue;
        for (int w = 0; w < std::max(1, n_warm_of[li] / 4); ++w) L[li].conv(&descs[li], st);
        HIP_OK(hipEventRecord(e0, st));
        for (int it = 0; it < n_time_of[li]; ++it) L[li].conv(&descs[li], st);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipEventSynchronize(e1));
        float ms;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        samples[li].push_back(ms * 1e3 / n_time_of[li]);
      }
    for (size_t li = 0; li < L.size(); ++li) {
      if (samples[li].empty()) continue;
      std::vector<double> v = samples[li];
      std::sort(v.begin(), v.end());
      const double us = v[(v.size() - 1) / 2];
      frame_us[li] += us * ly.calls;
      (ly.set == 1 ? big_us : small_us)[li] += us * ly.calls;
      if (csv) printf("\ncsv,%s,%zu,%.2f,%.3f,%.1f", ly.name, li, us, gf, ly.calls);
      else printf(" | %8.1f us %6.1f TF%s", us, gf / us * 1e3, tails[li].c_str());
    }
    printf("\n");
    fflush(stdout);
    for (void* p : keep) HIP_OK(hipFree(p));
  }
  printf("%-36s", "frame (sum calls x us), TF");
  for (size_t li = 0; li < L.size(); ++li) printf(" | %8.1f us %6.1f TF", frame_us[li], frame_gf / frame_us[li] * 1e3);
  printf("\n%-36s", "  big layers / small layers");
  for (size_t li = 0; li < L.size(); ++li) printf(" | %8.1f / %8.1f us ", big_us[li], small_us[li]);
  printf("\n");
  return 0;
}
