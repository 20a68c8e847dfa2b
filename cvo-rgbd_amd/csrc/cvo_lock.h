// cvo_lock.h -- one process-wide reader / writer lock around the library's use of the
// HIP runtime.
//
// Why: a stream capture (hipStreamBeginCapture .. EndCapture, how the loop's batches and
// the front end's frame become hipGraphs) is a process-wide affair in this runtime: while
// one host thread captures, plain launches / copies / allocations of ANOTHER thread on its
// own context and stream fail with "operation failed due to a previous error during
// capture", and they spoil the capture in turn.  Independent contexts must be usable from
// independent threads, so: every entry point that touches the runtime holds
// the lock shared, a capture window holds it exclusively.  Captures are rare (a few per
// context, then the graphs are cached); an entry point is re-entrant on its own thread.
#pragma once
#include <chrono>
#include <shared_mutex>

namespace cvo_lock {

inline std::shared_mutex &mutex()
{
    // (never destroyed: contexts may be closed from finalisers that run after static destructors)
    static std::shared_mutex *m = new std::shared_mutex;
    return *m;
}

inline int &depth()
{
    static thread_local int d = 0;
    return d;
}

// at the top of an entry point
struct Api {
    Api() { if (depth()++ == 0) mutex().lock_shared(); }
    ~Api() { if (--depth() == 0) mutex().unlock_shared(); }
    Api(const Api &) = delete;
    Api &operator=(const Api &) = delete;
};

// around a capture window, inside an entry point (which holds the lock shared).  The
// exclusive lock is only TRIED for a millisecond (readers may overlap without end when many
// threads are busy): `ok` false = no capture this time, the caller launches eagerly.
struct Capture {
    bool held, ok;
    Capture() : held(depth() > 0), ok(false)
    {
        if (held) mutex().unlock_shared();
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(1);
        do {
            ok = mutex().try_lock();
        } while (!ok && std::chrono::steady_clock::now() < deadline);
        if (!ok && held) mutex().lock_shared();
    }
    ~Capture()
    {
        if (!ok) return;
        mutex().unlock();
        if (held) mutex().lock_shared();
    }
    Capture(const Capture &) = delete;
    Capture &operator=(const Capture &) = delete;
};

}   // namespace cvo_lock
