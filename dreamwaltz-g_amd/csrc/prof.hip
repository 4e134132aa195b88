// prof.hip -- per-kernel HIP-event timing on the launch stream (include/dwg_prof.h).
#include "dwg_common.h"
#include "dwg_prof_internal.h"
#include "../../include/dwg_prof.h"

#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <cstdio>
#include <cstring>

namespace {
struct Sample { hipEvent_t a, b; std::string symbol; double work; };
std::mutex g_mu;
bool g_on = false;
// What an event bracket adds on top of the kernel it encloses: the two records themselves PLUS the dispatch of the kernel after the first
// record and its completion (release) before the second -- rocprofv3's kernel trace counts neither.  Measured once per enable on the first
// stream seen as the median bracket around a one-wave EMPTY kernel, minus 1 us for that kernel's own execution (rocprofv3 lists such launches
// at 1.0-1.2 us), and subtracted from every sample, so that the reported averages are kernel durations comparable with the kernel trace.
// Round 6: the bracket used to be calibrated WITHOUT a kernel inside (records only): 17.6 us launches read 21.0 us, and the symbol with
// the largest total came out differently from rocprofv3's.
float g_bracket_ms = -1.f;

__global__ void k_prof_empty() {}

float calibrate_bracket(hipStream_t stream) {
    const int N = 15;
    hipEvent_t a[N], b[N];
    float v[N];
    int n = 0;
    hipLaunchKernelGGL(k_prof_empty, dim3(1), dim3(64), 0, stream);        // (first launch: code object load)
    for (int i = 0; i < N; i++) {
        if (hipEventCreate(&a[i]) != hipSuccess || hipEventCreate(&b[i]) != hipSuccess) break;
        hipEventRecord(a[i], stream);
        hipLaunchKernelGGL(k_prof_empty, dim3(1), dim3(64), 0, stream);
        hipEventRecord(b[i], stream);
        n++;
    }
    if (n == 0) return 0.f;
    hipEventSynchronize(b[n - 1]);
    for (int i = 0; i < n; i++) { v[i] = 0.f; hipEventElapsedTime(&v[i], a[i], b[i]); hipEventDestroy(a[i]); hipEventDestroy(b[i]); }
    for (int i = 0; i < n; i++)
        for (int j = i + 1; j < n; j++) if (v[j] < v[i]) { float t = v[i]; v[i] = v[j]; v[j] = t; }
    const float med = v[n / 2] - 0.001f;
    return med > 0.f ? med : 0.f;
}
std::map<std::string, std::vector<Sample>> g_samples;

void clear_locked() {
    for (auto& kv : g_samples)
        for (auto& s : kv.second) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    g_samples.clear();
}
}  // namespace

bool dwg_prof_on() { return g_on; }

int& dwg_launch_failed_flag() {
    static thread_local int flag = 0;
    return flag;
}

void dwg_prof_begin(const char* name, const char* symbol, double work, hipStream_t stream, void** token) {
    *token = nullptr;
    if (!g_on) return;
    if (g_bracket_ms < 0.f) g_bracket_ms = calibrate_bracket(stream);
    Sample s;
    s.symbol = symbol ? symbol : name; s.work = work;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
    hipEventRecord(s.a, stream);
    std::lock_guard<std::mutex> lk(g_mu);
    auto& v = g_samples[name];
    v.push_back(s);
    *token = (void*)(uintptr_t)v.size();  // 1-based index
}

void dwg_prof_end(const char* name, hipStream_t stream, void* token) {
    if (!token) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_samples.find(name);
    if (it == g_samples.end()) return;
    size_t idx = (size_t)(uintptr_t)token - 1;
    if (idx < it->second.size()) hipEventRecord(it->second[idx].b, stream);
}

extern "C" {

int dwg_prof_enable(int32_t enable) {
    std::lock_guard<std::mutex> lk(g_mu);
    clear_locked();
    g_on = enable != 0;
    g_bracket_ms = -1.f;
    return DWG_OK;
}

int dwg_prof_query(const char* name, int64_t* count, double* total_ms) {
    if (!name || !count || !total_ms) return DWG_E_ARG;
    std::lock_guard<std::mutex> lk(g_mu);
    *count = 0; *total_ms = 0.0;
    auto it = g_samples.find(name);
    if (it == g_samples.end()) return DWG_OK;
    for (auto& s : it->second) {
        if (hipEventSynchronize(s.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { ms -= g_bracket_ms > 0.f ? g_bracket_ms : 0.f; *total_ms += ms > 0.f ? ms : 0.f; *count += 1; }
    }
    return DWG_OK;
}

int64_t dwg_prof_dump(char* buf, int64_t cap) {
    std::string out;
    std::vector<std::string> names;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_samples) names.push_back(kv.first);
    }
    for (auto& n : names) {
        int64_t c = 0; double ms = 0.0;
        dwg_prof_query(n.c_str(), &c, &ms);
        char line[256];
        snprintf(line, sizeof(line), "%s %lld %.6f\n", n.c_str(), (long long)c, ms);
        out += line;
    }
    if (buf && cap > 0) {
        size_t n = out.size() < (size_t)(cap - 1) ? out.size() : (size_t)(cap - 1);
        memcpy(buf, out.data(), n); buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

int64_t dwg_prof_dump_symbols(char* buf, int64_t cap) {
    struct Agg { long long n = 0; double ms = 0.0, work = 0.0; };
    std::map<std::string, Agg> agg;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto& kv : g_samples)
            for (auto& s : kv.second) {
                if (hipEventSynchronize(s.b) != hipSuccess) continue;
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
                ms -= g_bracket_ms > 0.f ? g_bracket_ms : 0.f;
                Agg& a = agg[s.symbol];
                a.n += 1; a.ms += ms > 0.f ? ms : 0.f; a.work += s.work;
            }
    }
    std::string out;
    for (auto& kv : agg) {
        char line[384];
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\t%.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.work);
        out += line;
    }
    if (buf && cap > 0) {
        size_t n = out.size() < (size_t)(cap - 1) ? out.size() : (size_t)(cap - 1);
        memcpy(buf, out.data(), n); buf[n] = 0;
    }
    return (int64_t)out.size() + 1;
}

}  // extern "C"
