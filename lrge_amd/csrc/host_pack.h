// host_pack.h -- the 2-bit pack (K0) on the HOST for read sets that start in host memory, and the background uploader.
//
// lrge_hip_seqset_upload* hands over ASCII bases (the (name, seq) records of twoset.rs:216-241).  Sent as they are they
// cross PCIe at 1 byte per base -- 13 ms for the 720 Mbases of the headline workload's targets, none of which can hide
// behind anything (the index needs all of them).  The packed image the device works on is 0.375 bytes per base, so a set
// that starts in host memory is packed by a few host threads (AVX2 + BMI2 where the CPU has them: ~40 instructions per 32
// bases) in chunks, chunk i travelling while chunk i + 1 is packed; the device-side k_pack (k_sketch.h) remains the path of
// reads that are already resident in HBM, and both produce the same image bit for bit (tests/test_gpu_upload.py).
// The work is done by an uploader thread per context, so that lrge_hip_seqset_upload_async returns at once and a second
// set is packed and travels while the first one is being indexed (the reference's producer thread, twoset.rs:216-241).
#pragma once
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "internal.h"

// A/a=0 C/c=1 G/g=2 T/t/U/u=3, everything else "ambiguous" (mask bit set, code 0): the host twin of nt4_code() in k_sketch.h
static inline u32 hp_nt4(u8 c) {
    const u32 b = (c >> 1) & 3;
    const bool letter = (c & 0xC0u) == 0x40u && ((0x0030008Au >> (c & 31)) & 1u);
    return letter ? (b ^ (b >> 1)) : 4u;
}

// one 32-base word: cnt bases at src (cnt <= 32; positions past cnt are padding: ambiguous, code 0)
static inline void hp_word_scalar(const u8 *src, u32 cnt, u64 *bits, u32 *mask) {
    u64 b = 0; u32 m = 0;
    for (u32 i = 0; i < cnt; ++i) { const u32 c = hp_nt4(src[i]); b |= (u64)(c & 3) << (2 * i); m |= (c >> 2) << i; }
    if (cnt < 32) m |= ~0u << cnt;
    *bits = b; *mask = m;
}

#if defined(__x86_64__)
// 32 bases at once.  Letters: a table lookup on the low nibble of the case-folded byte gives the high nibble the byte must have
// (A 0x41, C 0x43, G 0x47: 4; T 0x54, U 0x55: 5).  Codes: bits 1..2 of the byte, b ^ (b >> 1), zeroed for non-letters, then two
// multiply-adds put four 2-bit codes into the low byte of every dword (c0 + 4 c1, then x0 + 16 x1) and one byte shuffle
// collects them: ~18 vector operations per word, AVX2 only (the first form used four pext / pdep pairs per word, ~45 operations and
// BMI2).  MEASURED: the pack of the headline workload's 720 Mbases takes 6.1-6.3 ms on 32 threads either way, and 5.7 on 96 -- it
// reads the caller's buffer at ~120 GB/s, which is what ONE NUMA node of the host delivers: the memory, not the arithmetic.
__attribute__((target("avx2"))) static inline void hp_word_avx2(const u8 *src, u64 *bits, u32 *mask) {
    const __m256i v = _mm256_loadu_si256((const __m256i *)src);
    const __m256i up = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));            // fold case
    const __m256i nib = _mm256_set1_epi8(0x0F);
    const __m256i want = _mm256_setr_epi8(-1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1,
                                          -1, 4, -1, 4, 5, 5, -1, 4, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i hi = _mm256_and_si256(_mm256_srli_epi16(up, 4), nib);
    const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(want, _mm256_and_si256(up, nib)), hi);
    const u32 letters = (u32)_mm256_movemask_epi8(ok);
    __m256i c = _mm256_and_si256(_mm256_srli_epi16(v, 1), _mm256_set1_epi8(3));                    // bits 1..2 of every byte
    c = _mm256_xor_si256(c, _mm256_and_si256(_mm256_srli_epi16(c, 1), _mm256_set1_epi8(1)));      // b ^ (b >> 1)
    c = _mm256_and_si256(c, ok);
    const __m256i x = _mm256_maddubs_epi16(c, _mm256_set1_epi16(0x0401));                           // c0 + 4 c1 per 16 bits
    const __m256i y = _mm256_madd_epi16(x, _mm256_set1_epi32(0x00100001));                          // x0 + 16 x1 per 32 bits: 4 codes in the low byte
    const __m256i pick = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i z = _mm256_shuffle_epi8(y, pick);
    const u64 lo = (u32)_mm256_cvtsi256_si32(z), hi32 = (u32)_mm256_extract_epi32(z, 4);
    *bits = lo | hi32 << 32; *mask = ~letters;
}
#endif

static bool hp_have_avx2() {
#if defined(__x86_64__)
    static const bool ok = __builtin_cpu_supports("avx2");
    return ok;
#else
    return false;
#endif
}

// words [w0, w1) of the packed image of a set (reads start on word boundaries: woff[]; boff[] = base offsets of the reads
// inside `ascii`) into out_pack / out_mask (indexed from w0)
static void hp_pack_range(const u8 *ascii, const u64 *boff, const u64 *woff, u32 n_reads, u64 w0, u64 w1, u64 *out_pack, u32 *out_mask) {
    if (w0 >= w1) return;
    // the read that holds word w0: the last read whose first word is <= w0 and that owns at least one word
    u32 r = (u32)(std::upper_bound(woff, woff + n_reads + 1, w0) - woff) - 1;
    const bool simd = hp_have_avx2();
    for (u64 w = w0; w < w1;) {
        while (r + 1 < n_reads && woff[r + 1] <= w) ++r;
        const u64 len = boff[r + 1] - boff[r], wend = std::min(w1, woff[r + 1]);
        const u8 *base = ascii + boff[r];
        for (; w < wend; ++w) {
            const u64 pos = (w - woff[r]) * 32;
            const u32 cnt = (u32)std::min<u64>(32, len - pos);
            u64 b; u32 m;
#if defined(__x86_64__)
            if (simd && cnt == 32) hp_word_avx2(base + pos, &b, &m); else
#endif
            hp_word_scalar(base + pos, cnt, &b, &m);
            out_pack[w - w0] = b; out_mask[w - w0] = m;
        }
    }
}

// The CPUs of the NUMA node the GPU hangs off (sysfs: the PCI device's numa_node and that node's cpulist), cut down to the
// process's own affinity mask; empty when the system does not say.  Pinned host memory lives on that node whoever touches it
// first, and the pack runs at twice the speed from its cores (measured on a two-socket box: 6.6 ms against 12.1 ms for 720
// Mbases, whichever node the source had been filled from: tools/micro/upload_numa.py) -- left to the scheduler a process ends up
// anywhere in between, and stays there.
static std::vector<int> hp_gpu_node_cpus(int device, int *node_out = nullptr) {
    std::vector<int> out;
    if (node_out) *node_out = -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return out; }
    for (char *c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return out;
    if (node_out) *node_out = node;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return out;
    cpu_set_t mine; CPU_ZERO(&mine);
    const bool have_mine = sched_getaffinity(0, sizeof(mine), &mine) == 0;
    int a, b; char sep;
    while (fscanf(f, "%d", &a) == 1) {
        b = a;
        if (fscanf(f, "%c", &sep) == 1 && sep == '-') { if (fscanf(f, "%d", &b) != 1) b = a; if (fscanf(f, "%c", &sep) != 1) sep = 0; }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (!have_mine || CPU_ISSET(c, &mine)) out.push_back(c);
        if (sep != ',') break;
    }
    fclose(f);
    return out;
}
// CPUs this process may keep busy at once: the smaller of its affinity mask and its cgroup's CPU bandwidth limit (v2 cpu.max, v1
// cpu.cfs_quota_us / cpu.cfs_period_us).  The GPU boxes of this pool show 256 hardware threads and grant 16 CPUs' worth of time
// (cpu.max = "1600000 100000"): 32 pack threads -- hardware_concurrency / 2, clamped -- beside a main thread that launches kernels spent
// part of every period throttled (round 5: the pack pool is sized by THIS figure, and one pool serves every context of the process).
static double hp_cpu_quota() {
    double q = (double)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t mine; CPU_ZERO(&mine);
    if (sched_getaffinity(0, sizeof(mine), &mine) == 0 && CPU_COUNT(&mine) > 0) q = std::min(q, (double)CPU_COUNT(&mine));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[32] = {0}; long per = 0;
        if (fscanf(f, "%31s %ld", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0 && atof(a) > 0) q = std::min(q, atof(a) / (double)per);
        fclose(f);
    } else {
        long quota = -1, per = 0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &per) != 1) per = 0; fclose(g); }
        if (quota > 0 && per > 0) q = std::min(q, (double)quota / (double)per);
    }
    return std::max(1.0, q);
}

static void hp_pin_thread(std::thread &t, const std::vector<int> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t set; CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);
}

// A few persistent host threads: parallel_for(n, fn) runs fn(0) .. fn(n - 1) on them (and on the caller).
struct HostPool {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    std::function<void(u32)> fn; u32 n_tasks = 0; std::atomic<u32> next{0}; u32 running = 0; u64 gen = 0; bool quit = false;
    void start(u32 n_threads, const std::vector<int> &cpus = std::vector<int>()) {
        if (!th.empty()) return;
        for (u32 t = 0; t < n_threads; ++t) { th.emplace_back([this] { worker(); }); hp_pin_thread(th.back(), cpus); }
    }
    void worker() {
        u64 seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv_go.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
            for (u32 i; (i = next.fetch_add(1)) < n_tasks;) fn(i);
            { std::lock_guard<std::mutex> lk(mu); if (--running == 0) cv_done.notify_all(); }
        }
    }
    std::mutex use_mu;      // one parallel_for at a time: the pool is shared by the uploaders of every context of the process
    void parallel_for(u32 n, std::function<void(u32)> f) {
        if (n == 0) return;
        if (th.empty() || n == 1) { for (u32 i = 0; i < n; ++i) f(i); return; }
        std::lock_guard<std::mutex> use(use_mu);
        { std::lock_guard<std::mutex> lk(mu); fn = std::move(f); n_tasks = n; next = 0; running = (u32)th.size(); ++gen; }
        cv_go.notify_all();
        for (u32 i; (i = next.fetch_add(1)) < n_tasks;) fn(i);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return running == 0; });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

// ONE pack pool per process and NUMA node (round 5): eight contexts of one process -- the ranks of a local communicator, a world
// emulated on one GPU -- used to start min(32, hw / 2) workers EACH, 256 pack threads on a host that grants 16 CPUs; now they take
// turns, chunk by chunk, on a pool sized for the host.  The pool lives as long as the process.
static HostPool *hp_shared_pool(int node, u32 n_threads, const std::vector<int> &cpus) {
    static std::mutex mu; static std::map<int, HostPool *> pools;
    std::lock_guard<std::mutex> g(mu);
    HostPool *&p = pools[node];
    if (!p) { p = new HostPool(); p->start(n_threads, cpus); }
    return p;
}

// The uploader of a context: jobs run one after the other on its thread (FIFO: a second set is packed behind the first).
struct Uploader {
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::deque<std::function<void()>> jobs; bool quit = false, started = false;
    HostPool *pool = nullptr;           // hp_shared_pool: shared with the other contexts of the process on the same NUMA node
    int node = -1;
    std::vector<int> cpus;              // where the uploader and its workers run (hp_gpu_node_cpus; empty: wherever)
    void submit(std::function<void()> j) {
        std::lock_guard<std::mutex> lk(mu);
        if (!started) { started = true; th = std::thread([this] { run(); }); hp_pin_thread(th, cpus); }
        jobs.push_back(std::move(j));
        cv.notify_all();
    }
    bool busy = false;                  // a job is running on the uploader thread
    void run() {
        for (;;) {
            std::function<void()> j;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return quit || !jobs.empty(); }); if (jobs.empty()) return; j = std::move(jobs.front()); jobs.pop_front(); busy = true; }
            j();
            { std::lock_guard<std::mutex> lk(mu); busy = false; }
            cv.notify_all();
        }
    }
    // every queued job has run to its end (the staging buffers may be replaced: ADVICE r03)
    void drain() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return jobs.empty() && !busy; }); }
    ~Uploader() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// completion of one upload job (the set's side of it)
struct UploadJob {
    std::mutex mu; std::condition_variable cv; bool done = false; int rc = 0; std::string err;
    // chunk gates: chunk j of the packed image -- words [.., gate_w1[j]) -- is on its way once gate_ev[j] has been RECORDED on the
    // copy stream (gates_recorded > j); a consumer that waits for that event on the device may work on those words while
    // the later chunks are still being packed (the index sketch does: host_sketch.inl, sketch_launch)
    std::vector<hipEvent_t> gate_ev; std::vector<u64> gate_w1; int gates_recorded = 0;
    void gate_recorded() { { std::lock_guard<std::mutex> lk(mu); ++gates_recorded; } cv.notify_all(); }
    // true when gate j has been recorded; false when the job ended (failed) before it
    bool wait_gate(int j) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return gates_recorded > j || done; }); return gates_recorded > j; }
    void finish(int r, const std::string &e) { { std::lock_guard<std::mutex> lk(mu); done = true; rc = r; err = e; } cv.notify_all(); }
    bool is_done() { std::lock_guard<std::mutex> lk(mu); return done; }
    int wait(std::string *e) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); if (rc && e) *e = err; return rc; }
};
