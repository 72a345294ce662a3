// Host-side runtime bits of libapex_mi355.so: error text, launch checks and the HIP-event
// kernel timer behind apexmi_prof_* (bench.py's roofline leg reads it).
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
thread_local char g_err[512] = "";

struct ProfRec {
    int cls;
    hipEvent_t start, stop;
    double flops, bytes;
};
bool g_prof_on = false;
std::mutex g_prof_mu;
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

void apexmi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int apexmi_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        apexmi_set_error("%s: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

ApexmiProfScope::ApexmiProfScope(int cls_, hipStream_t s, double flops, double bytes)
    : cls(cls_), stream(s), slot(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.cls = cls;
    r.start = get_event();
    r.stop = get_event();
    r.flops = flops;
    r.bytes = bytes;
    (void)hipEventRecord(r.start, stream);
    slot = (int)g_recs.size();
    g_recs.push_back(r);
}

ApexmiProfScope::~ApexmiProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_recs[slot].stop, stream);
}

extern "C" int apexmi_version(void) { return 100; }  // 0.1.0

extern "C" const char* apexmi_last_error(void) { return g_err; }

extern "C" int apexmi_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

extern "C" int apexmi_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) {
        g_pool.push_back(r.start);
        g_pool.push_back(r.stop);
    }
    g_recs.clear();
    return 0;
}

extern "C" int apexmi_prof_read(double ms[APEXMI_NCLASS], int64_t launches[APEXMI_NCLASS],
                                double flops[APEXMI_NCLASS], double bytes[APEXMI_NCLASS]) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        apexmi_set_error("prof_read: %s", hipGetErrorString(e));
        return 2;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int c = 0; c < APEXMI_NCLASS; ++c) {
        ms[c] = 0;
        launches[c] = 0;
        flops[c] = 0;
        bytes[c] = 0;
    }
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) continue;
        ms[r.cls] += t;
        launches[r.cls] += 1;
        flops[r.cls] += r.flops;
        bytes[r.cls] += r.bytes;
    }
    return 0;
}

// ---- live clock probe: shader cycles and 100 MHz reference ticks summed over the GEMM workgroups' K-loops -------------------
namespace {
bool g_clk_on = false;
unsigned long long* g_clk_dev[64] = {};
}  // namespace
unsigned long long* apexmi_clk_ptr() {
    if (!g_clk_on) return nullptr;
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < 64 ? g_clk_dev[dev] : nullptr;
}
extern "C" int apexmi_clk_enable(int on) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) {
        apexmi_set_error("clk_enable: device ordinal %d out of range", dev);
        return 1;
    }
    if (on && g_clk_dev[dev] == nullptr) {
        hipError_t e = hipMalloc((void**)&g_clk_dev[dev], 2 * sizeof(unsigned long long));
        if (e != hipSuccess) {
            apexmi_set_error("clk_enable: %s", hipGetErrorString(e));
            return 1;
        }
    }
    if (on) (void)hipMemset(g_clk_dev[dev], 0, 2 * sizeof(unsigned long long));
    g_clk_on = on != 0;
    return 0;
}
extern "C" int apexmi_clk_read(uint64_t* cycles, uint64_t* ref_ticks) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!cycles || !ref_ticks || dev < 0 || dev >= 64 || g_clk_dev[dev] == nullptr) {
        apexmi_set_error("clk_read: the probe was never enabled on this device");
        return 1;
    }
    unsigned long long h[2] = {0, 0};
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h, g_clk_dev[dev], sizeof(h), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        apexmi_set_error("clk_read: %s", hipGetErrorString(e));
        return 1;
    }
    *cycles = h[0];
    *ref_ticks = h[1];
    return 0;
}

// ---- tuning switches (A/B levers; defaults are the shipped configuration, include/apexmi.h lists the keys) ----
void apexmi_set_attn_waves(int v);
void apexmi_set_attn_mfma(int v);
void apexmi_set_ln_wave(int v);
void apexmi_set_attn_c4(int v);
void apexmi_set_attn_stages(int v);
void apexmi_set_attn_xv(int v);
void apexmi_set_attn_w64(int v);
void apexmi_set_qk_group(int v);
void apexmi_set_attn_split(int v);
void apexmi_set_conv_v2(int v);
void apexmi_set_conv_slab(int v);
void apexmi_set_conv_pp(int v);
void apexmi_set_conv_dbg(int v);
void apexmi_set_conv_torder(int v);
void apexmi_set_conv_prof(int half, int v);
int apexmi_set_gemm_key(const char* key, int value);

extern "C" int apexmi_tune_set(const char* key, int value) {
    if (!key) {
        apexmi_set_error("tune_set: null key");
        return 1;
    }
    if (!strncmp(key, "gemm.", 5)) {
        if (apexmi_set_gemm_key(key, value) == 0) return 0;
    } else if (!strcmp(key, "conv.v2")) {
        apexmi_set_conv_v2(value);
        return 0;
    } else if (!strcmp(key, "conv.slab")) {
        apexmi_set_conv_slab(value);
        return 0;
#ifdef APEXMI_DEBUG   // experiment knobs of tools/conv_prof.py / conv_ablate.py: `conv.dbg` skips DMA (WRONG results) and
                      // `conv.prof_*` makes the kernels write cycle stamps through a caller-supplied pointer — never in a release build
    } else if (!strcmp(key, "conv.prof_lo") || !strcmp(key, "conv.prof_hi")) {
        apexmi_set_conv_prof(key[10] == 'h', value);
        return 0;
    } else if (!strcmp(key, "conv.dbg")) {
        apexmi_set_conv_dbg(value);
        return 0;
#else
    } else if (!strcmp(key, "conv.prof_lo") || !strcmp(key, "conv.prof_hi") || !strcmp(key, "conv.dbg")) {
        apexmi_set_error("tune_set: '%s' is a debug knob; rebuild with APEXMI_DEBUG=1 (python -m apex_studio_amd.build)", key);
        return 1;
#endif
    } else if (!strcmp(key, "conv.torder")) {
        apexmi_set_conv_torder(value);
        return 0;
    } else if (!strcmp(key, "conv.pp")) {
        apexmi_set_conv_pp(value);
        return 0;
    } else if (!strcmp(key, "attn.split")) {
        apexmi_set_attn_split(value);
        return 0;
    } else if (!strcmp(key, "qk.group")) {
        apexmi_set_qk_group(value);
        return 0;
    } else if (!strcmp(key, "attn.stages")) {
        apexmi_set_attn_stages(value);
        return 0;
    } else if (!strcmp(key, "attn.w64")) {
        apexmi_set_attn_w64(value);
        return 0;
    } else if (!strcmp(key, "attn.xv")) {
        apexmi_set_attn_xv(value);
        return 0;
    } else if (!strcmp(key, "attn.waves")) {
        apexmi_set_attn_waves(value);
        return 0;
    } else if (!strcmp(key, "attn.c4")) {
        apexmi_set_attn_c4(value);
        return 0;
    } else if (!strcmp(key, "ln.wave")) {
        apexmi_set_ln_wave(value);
        return 0;
    } else if (!strcmp(key, "attn.mfma")) {
        apexmi_set_attn_mfma(value);
        return 0;
    }
    apexmi_set_error("tune_set: unknown key");
    return 1;
}
