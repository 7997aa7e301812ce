// prof.cpp -- see prof.h
#include "prof.h"

#include <vector>

namespace r3g {
namespace {
struct Rec { hipEvent_t a, b; int cat; double work, bytes; };
constexpr int kRing = 2048;
bool g_on = false;
std::vector<Rec> g_ring;
int g_used = 0;
long long g_cnt[PC_COUNT];
double g_ms[PC_COUNT], g_work[PC_COUNT], g_bytes[PC_COUNT];

void drain() {
    if (g_used == 0) return;
    (void)hipEventSynchronize(g_ring[g_used - 1].b);
    for (int i = 0; i < g_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ring[i].a, g_ring[i].b) == hipSuccess) {
            g_cnt[g_ring[i].cat] += 1;
            g_ms[g_ring[i].cat] += ms;
            g_work[g_ring[i].cat] += g_ring[i].work;
            g_bytes[g_ring[i].cat] += g_ring[i].bytes;
        }
    }
    g_used = 0;
}
}  // namespace

void prof_enable(bool on) {
    if (on) {
        if (g_ring.empty()) {
            g_ring.resize(kRing);
            for (auto& r : g_ring) {
                (void)hipEventCreate(&r.a);
                (void)hipEventCreate(&r.b);
            }
        }
        g_used = 0;
        for (int i = 0; i < PC_COUNT; ++i) { g_cnt[i] = 0; g_ms[i] = 0; g_work[i] = 0; g_bytes[i] = 0; }
    } else {
        drain();
    }
    g_on = on;
}

bool prof_enabled() { return g_on; }

ProfScope::ProfScope(int cat, double work, hipStream_t stream, double bytes) : slot(-1), s(stream) {
    if (!g_on) return;
    if (g_used == kRing) drain();
    slot = g_used++;
    g_ring[slot].cat = cat;
    g_ring[slot].work = work;
    g_ring[slot].bytes = bytes;
    (void)hipEventRecord(g_ring[slot].a, s);
}

ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_ring[slot].b, s);
}

void prof_read(long long* counts, double* ms, double* work) {
    drain();
    for (int i = 0; i < PC_COUNT; ++i) {
        counts[i] = g_cnt[i];
        ms[i] = g_ms[i];
        work[i] = g_work[i];
    }
}

void prof_read_bytes(double* bytes) {
    drain();
    for (int i = 0; i < PC_COUNT; ++i) bytes[i] = g_bytes[i];
}

}  // namespace r3g
