// How to get a large pageable host buffer to the device fastest: plain hipMemcpy, hipHostRegister + copy,
// or several threads staging through pinned buffers.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 360u << 20;
    std::vector<char> host(n, 1);
    char *dev;
    (void)hipMalloc(&dev, n);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        (void)hipMemcpy(dev, host.data(), n, hipMemcpyHostToDevice);
        double t1 = now();
        printf("hipMemcpy pageable: %.1f ms (%.1f GB/s)\n", 1e3 * (t1 - t0), n / (t1 - t0) / 1e9);
        t0 = now();
        (void)hipHostRegister(host.data(), n, hipHostRegisterDefault);
        double tr = now();
        (void)hipMemcpy(dev, host.data(), n, hipMemcpyHostToDevice);
        double tc = now();
        (void)hipHostUnregister(host.data());
        t1 = now();
        printf("register %.1f ms + copy %.1f ms + unregister %.1f ms = %.1f ms\n", 1e3 * (tr - t0), 1e3 * (tc - tr), 1e3 * (t1 - tc), 1e3 * (t1 - t0));
        // staged: T threads, each with two pinned 8 MB buffers and its own stream
        for (int T : {4, 8, 16}) {
            const size_t chunk = 8u << 20;
            std::vector<char *> pin(2 * T);
            std::vector<hipStream_t> st(T);
            for (auto &p : pin) (void)hipHostMalloc(&p, chunk);
            for (auto &s : st) (void)hipStreamCreate(&s);
            t0 = now();
            std::vector<std::thread> th;
            const size_t n_chunks = (n + chunk - 1) / chunk;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    int flip = 0;
                    hipEvent_t ev[2];
                    (void)hipEventCreate(&ev[0]); (void)hipEventCreate(&ev[1]);
                    bool used[2] = {false, false};
                    for (size_t c = t; c < n_chunks; c += T) {
                        const size_t off = c * chunk, len = std::min(chunk, n - off);
                        if (used[flip]) (void)hipEventSynchronize(ev[flip]);
                        memcpy(pin[2 * t + flip], host.data() + off, len);
                        (void)hipMemcpyAsync(dev + off, pin[2 * t + flip], len, hipMemcpyHostToDevice, st[t]);
                        (void)hipEventRecord(ev[flip], st[t]);
                        used[flip] = true;
                        flip ^= 1;
                    }
                    (void)hipStreamSynchronize(st[t]);
                });
            for (auto &x : th) x.join();
            t1 = now();
            printf("staged, %2d threads: %.1f ms (%.1f GB/s)\n", T, 1e3 * (t1 - t0), n / (t1 - t0) / 1e9);
            for (auto &p : pin) (void)hipHostFree(p);
            for (auto &s : st) (void)hipStreamDestroy(s);
        }
    }
    return 0;
}
