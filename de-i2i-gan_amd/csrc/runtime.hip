// Library bookkeeping: device query, error strings, in-library HIP-event timing of one kernel family.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/dei2i_hip.h"
#include "launch.h"

namespace dei2i {

extern int g_v2_ablate;
extern int g_use_wgrad_v2;
extern int g_use_wgrad_halo;
extern int g_wgrad_halo_cbw, g_wgrad_halo_abl;
extern int g_use_wgrad_thin;
extern int g_dgrad_s2_ring;
extern int g_wt_splits_per_cu;
extern int g_halo_mfma32;
extern int g_halo_bn, g_halo_stages;
extern int g_halo16, g_halo16_stages, g_halo16_fold, g_halo16_s2;
extern unsigned long long* g_v2_dbg;
static int g_cus = 256;
void set_num_cu_rt(int n) { g_cus = n > 0 ? n : 256; }
int num_cu() { return g_cus; }

struct ProfState {
  bool on = false;
  bool events = true;          // false: count launches and FLOPs only (no HIP events: nothing is added to the stream)
  int every = 1;               // events around every `every`-th launch only (a sample: two event records break back-to-back dispatch)
  bool armed = false;          // the current launch is bracketed
  double flops_timed = 0.0;    // FLOPs of the bracketed launches
  size_t counted = 0;
  std::vector<hipEvent_t> starts, stops;
  size_t used = 0;
  double flops = 0.0;
};
static ProfState g_prof[PROF_FAMILIES];

static long long g_launches[K_COUNT];
void count_launch(int kid) { g_launches[kid]++; }
static const char* const g_kernel_names[K_COUNT] = {"gather_v1", "gather_v2", "halo_conv", "halo_conv_fp8", "thin_cin", "thin_cout",
                                                    "wgrad_v1", "wgrad_v2", "wgrad_halo", "wgrad_thin", "halo16_conv", "splitk_finalize", "halo16_s2"};

void prof_begin(int family, double flops, hipStream_t st) {
  ProfState& p = g_prof[family];
  if (!p.on) return;
  p.counted++;
  p.flops += flops;
  p.armed = p.events && (p.counted - 1) % (size_t)p.every == 0;
  if (!p.armed) return;
  p.flops_timed += flops;
  if (p.used == p.starts.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    p.starts.push_back(a);
    p.stops.push_back(b);
  }
  hipEventRecord(p.starts[p.used], st);
}

void prof_end(int family, hipStream_t st) {
  ProfState& p = g_prof[family];
  if (!p.on || !p.armed) return;
  hipEventRecord(p.stops[p.used], st);
  p.used++;
}

}  // namespace dei2i

using namespace dei2i;

extern "C" {

int dei2i_version(void) { return 100; }

int dei2i_init(int device) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return (int)e;
  set_num_cu_rt(prop.multiProcessorCount);
  set_num_cu(prop.multiProcessorCount);
  return 0;
}

const char* dei2i_error_string(int code) {
  if (code == DEI2I_ERR_BAD_ARG) return "dei2i: bad argument";
  if (code == DEI2I_ERR_WORKSPACE) return "dei2i: workspace too small";
  if (code >= 0) return hipGetErrorString((hipError_t)code);
  return "dei2i: unknown error";
}

int dei2i_set_option(const char* name, int value) {
  if (name == nullptr) return DEI2I_ERR_BAD_ARG;
  if (std::string(name) == "gather_gemm_v2") { set_use_v2(value); return 0; }
  if (std::string(name) == "wgrad_v2") { g_use_wgrad_v2 = value; return 0; }
  if (std::string(name) == "halo_conv") { set_use_halo(value); return 0; }
  if (std::string(name) == "thin_conv") { set_use_thin(value); return 0; }
  if (std::string(name) == "halo_mfma32") { g_halo_mfma32 = value; return 0; }
  if (std::string(name) == "halo_bn") { if (value != 0 && value != 64 && value != 128) return DEI2I_ERR_BAD_ARG; g_halo_bn = value; return 0; }
  if (std::string(name) == "halo_stages") {
    if (value != 0 && value != 4 && value != 6 && value != 8) return DEI2I_ERR_BAD_ARG;
    g_halo_stages = value;
    return 0;
  }
  if (std::string(name) == "halo16") { g_halo16 = value; return 0; }
  if (std::string(name) == "halo16_fold") { g_halo16_fold = value; return 0; }
  if (std::string(name) == "halo16_s2") { g_halo16_s2 = value; return 0; }
  if (std::string(name) == "halo16_stages") {
    if (value != 4 && value != 6 && value != 8) return DEI2I_ERR_BAD_ARG;
    g_halo16_stages = value;
    return 0;
  }
  if (std::string(name) == "splitk_atomic") { set_splitk_atomic(value); return 0; }
  if (std::string(name) == "wgrad_halo") { g_use_wgrad_halo = value; return 0; }
  if (std::string(name) == "wgrad_halo_abl") { g_wgrad_halo_abl = value; return 0; }
  if (std::string(name) == "wgrad_halo_cbw") { g_wgrad_halo_cbw = value == 1 ? 1 : 2; return 0; }
  if (std::string(name) == "wgrad_thin") { g_use_wgrad_thin = value; return 0; }
  if (std::string(name) == "dgrad_s2_ring") { g_dgrad_s2_ring = value; return 0; }
  if (std::string(name) == "wgrad_thin_splits") { g_wt_splits_per_cu = value; return 0; }
  if (std::string(name) == "v2_ablate") { g_v2_ablate = value; return 0; }     // timing-only builds: 1 = no loads, 2 = no MFMA
  return DEI2I_ERR_BAD_ARG;
}

int dei2i_set_debug_buffer(void* p) { g_v2_dbg = (unsigned long long*)p; return 0; }

int dei2i_launch_counts(int64_t* out, int n) {
  for (int i = 0; i < n && i < K_COUNT; ++i) out[i] = (int64_t)g_launches[i];
  return K_COUNT;
}
void dei2i_launch_counts_reset(void) { for (int i = 0; i < K_COUNT; ++i) g_launches[i] = 0; }
const char* dei2i_kernel_name(int kid) { return kid >= 0 && kid < K_COUNT ? g_kernel_names[kid] : nullptr; }

int dei2i_prof_enable(int family, int on) {
  if (family < 0 || family >= PROF_FAMILIES) return DEI2I_ERR_BAD_ARG;
  ProfState& p = g_prof[family];
  p.on = on != 0;
  p.events = on != 2;
  p.every = on >= 3 ? on : 1;
  p.used = 0;
  p.counted = 0;
  p.flops = 0.0;
  p.flops_timed = 0.0;
  return 0;
}

int dei2i_prof_collect_timed(int family, int64_t* timed_launches, double* timed_flops) {
  if (family < 0 || family >= PROF_FAMILIES) return DEI2I_ERR_BAD_ARG;
  if (timed_launches) *timed_launches = (int64_t)g_prof[family].used;
  if (timed_flops) *timed_flops = g_prof[family].flops_timed;
  return 0;
}

int dei2i_prof_collect(int family, int64_t* launches, double* total_ms, double* total_flops) {
  if (family < 0 || family >= PROF_FAMILIES) return DEI2I_ERR_BAD_ARG;
  ProfState& p = g_prof[family];
  double ms = 0.0;
  for (size_t i = 0; i < p.used; ++i) {
    hipError_t e = hipEventSynchronize(p.stops[i]);
    if (e != hipSuccess) return (int)e;
    float t = 0.f;
    e = hipEventElapsedTime(&t, p.starts[i], p.stops[i]);
    if (e != hipSuccess) return (int)e;
    ms += t;
  }
  if (launches) *launches = (int64_t)p.counted;
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = p.flops;
  p.used = 0;
  p.counted = 0;
  p.flops = 0.0;
  p.flops_timed = 0.0;
  return 0;
}

}  // extern "C"
