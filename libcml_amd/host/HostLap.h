// HostLap.h — development clock of the host mirror.  CMLHOST_TIMING=1 prints the host clock at the named points of a call (stderr);
// CMLHOST_TIMING=sum keeps, per (call, point), the time since the previous point and prints the table when the process ends (no output inside the calls).
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

namespace cml_amd {

struct HostLapTable {
    struct Row { double us = 0; long calls = 0; };
    std::map<std::string, Row> rows;
    ~HostLapTable() {
        for (const auto& kv : rows) fprintf(stderr, "[host laps] %-52s calls %5ld  mean %8.1f us  total %9.1f us\n", kv.first.c_str(), kv.second.calls, kv.second.us / (double)kv.second.calls, kv.second.us);
    }
    static HostLapTable& get() { static HostLapTable t; return t; }
};

struct HostLap {
    const char* fn; std::chrono::steady_clock::time_point t0, tl; int mode;       // 0 off, 1 print, 2 sum
    static int modeOf() { const char* e = getenv("CMLHOST_TIMING"); return !e ? 0 : (strcmp(e, "sum") == 0 ? 2 : 1); }
    explicit HostLap(const char* f) : fn(f), t0(std::chrono::steady_clock::now()), tl(t0), mode(modeOf()) {}
    void operator()(const char* what) {
        if (!mode) return;
        const auto now = std::chrono::steady_clock::now();
        if (mode == 1) fprintf(stderr, "      [%s] %-22s %.0f us\n", fn, what, std::chrono::duration<double, std::micro>(now - t0).count());
        else { HostLapTable::Row& r = HostLapTable::get().rows[std::string(fn) + ": " + what]; r.us += std::chrono::duration<double, std::micro>(now - tl).count(); r.calls++; tl = std::chrono::steady_clock::now(); }
    }
};

}  // namespace cml_amd
