// host_util.hpp — small host-side utilities of the C ABI layer: the error type that carries a
// KBO_E_* code across internal calls, RAII device / pinned buffers, and the helper-thread team
// used for staging copies and offset scans.
#pragma once
#include "../../include/kbo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "refine.hpp"

namespace kbo_host {

inline std::string &last_error()
{
    thread_local std::string err;
    return err;
}

struct KboError : std::runtime_error {
    int code;
    KboError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define KBO_REQUIRE(cond, code, msg)                                                              \
    do {                                                                                          \
        if (!(cond)) throw KboError((code), (msg));                                               \
    } while (0)

#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            throw KboError(KBO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));        \
    } while (0)

template <typename F> inline int guarded(F f)
{
    try {
        f();
        return KBO_OK;
    } catch (const KboError &e) {
        last_error() = e.what();
        return e.code;
    } catch (const kbo::RefPanic &e) {
        last_error() = e.what();
        return KBO_E_REF_PANIC;
    } catch (const std::bad_alloc &) {
        last_error() = "out of host memory";
        return KBO_E_NOMEM;
    } catch (const std::exception &e) {
        last_error() = e.what();
        return KBO_E_BAD_ARG;
    }
}

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    explicit DevBuf(size_t bytes) { alloc(bytes); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    int dev = -1; // HIP device the memory lives on (the current one at allocation)
    void alloc(size_t bytes)
    {
        release();
        (void)hipGetDevice(&dev);
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) {
            p = nullptr;
            throw KboError(e == hipErrorOutOfMemory ? KBO_E_NOMEM : KBO_E_HIP,
                           std::string("hipMalloc: ") + hipGetErrorString(e));
        }
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t cap = 0;
    // grow-only: slot buffers are reused from slab to slab.  A buffer kept by a host thread across calls (thread_local
    // caches) is only reused on the device it was allocated on: a thread that has moved to another device gets fresh memory.
    void ensure(size_t bytes)
    {
        if (bytes <= cap && p) {
            int cur = -1;
            if (hipGetDevice(&cur) == hipSuccess && cur == dev) return;
        }
        alloc(bytes);
        cap = std::max<size_t>(bytes, 16);
    }
    ~DevBuf() { release(); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct PinBuf { // pinned host staging memory, grow-only
    void *p = nullptr;
    size_t cap = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    void ensure(size_t bytes)
    {
        if (bytes <= cap && p) return;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipHostMalloc(&p, std::max<size_t>(bytes, 16), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            throw KboError(KBO_E_NOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
        }
        cap = std::max<size_t>(bytes, 16);
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

// ---- host helper threads: staging copies and offset scans of the host batch entry points.
// A process-wide team (leaked on purpose: its threads sleep on a condition variable until the
// process ends); the calling thread takes part; one job at a time.
class HostTeam {
public:
    static HostTeam &get() // staging copies into pinned memory, offset scans
    {
        static HostTeam *team = new HostTeam();
        return *team;
    }
    static HostTeam &out() // copies out of pinned memory (runs next to get(): a second set of threads)
    {
        static HostTeam *team = new HostTeam();
        return *team;
    }
    void set_threads(unsigned n) { want_ = std::max(1u, std::min(n, 64u)); }
    // runs fn(0..n_tasks-1), returns when all are done
    void run(size_t n_tasks, const std::function<void(size_t)> &fn)
    {
        if (n_tasks == 0) return;
        if (n_tasks == 1 || want_ <= 1) { // (exceptions propagate as they are)
            for (size_t i = 0; i < n_tasks; i++) fn(i);
            return;
        }
        std::lock_guard<std::mutex> one_job(job_mu_);
        {
            std::lock_guard<std::mutex> g(mu_);
            while (threads_.size() + 1 < want_) threads_.emplace_back([this] { loop(); });
            fn_ = &fn;
            next_ = 0;
            total_ = n_tasks;
            remaining_ = n_tasks;
            gen_++;
        }
        cv_.notify_all();
        work();
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> g(mu_);
            done_cv_.wait(g, [&] { return remaining_ == 0; });
            fn_ = nullptr;
            std::swap(err, err_);
        }
        if (err) std::rethrow_exception(err); // the first exception a task threw, once every task has drained
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t piece = 2u << 20;
        run((bytes + piece - 1) / piece, [&](size_t i) {
            const size_t a = i * piece, b = std::min(bytes, a + piece);
            std::memcpy(static_cast<char *>(dst) + a, static_cast<const char *>(src) + a, b - a);
        });
    }

private:
    void work()
    {
        for (;;) {
            size_t i;
            const std::function<void(size_t)> *fn;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!fn_ || next_ >= total_) return;
                i = next_++;
                fn = fn_;
            }
            std::exception_ptr err;
            try {
                (*fn)(i);
            } catch (...) { // (on a helper thread an escaping exception would terminate the process)
                err = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                if (err && !err_) err_ = err;
                if (--remaining_ == 0) done_cv_.notify_all();
            }
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
            }
            work();
        }
    }
    std::mutex job_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> threads_;
    const std::function<void(size_t)> *fn_ = nullptr;
    std::exception_ptr err_;
    size_t next_ = 0, total_ = 0, remaining_ = 0;
    uint64_t gen_ = 0;
    unsigned want_ = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
};

} // namespace kbo_host
