/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 * tls_glue.cpp: a stand-in for the reference's src/libcore/tls.cpp (thread-local storage bookkeeping built on
 * boost::multi_index, which is not available here) behind the same interface (include/mitsuba/core/tls.h:30-60).
 * Infrastructure only: nothing on the rendering path is implemented here.
 */
#include <mitsuba/mitsuba.h>
#include <mitsuba/core/tls.h>
#include <mutex>
#include <set>
#include <thread>
#include <unordered_map>

MTS_NAMESPACE_BEGIN
namespace detail {

struct Store {
    ThreadLocalBase::ConstructFunctor construct;
    ThreadLocalBase::DestructFunctor destruct;
    std::mutex mutex;
    std::unordered_map<std::thread::id, void *> values;
};
struct ThreadLocalBase::ThreadLocalPrivate : Store { };

static std::mutex &registryLock() { static std::mutex m; return m; }
static std::set<Store *> &registry() { static std::set<Store *> s; return s; }

ThreadLocalBase::ThreadLocalBase(const ConstructFunctor &constructFunctor, const DestructFunctor &destructFunctor)
    : d(new ThreadLocalPrivate()) {
    d->construct = constructFunctor; d->destruct = destructFunctor;
    std::lock_guard<std::mutex> g(registryLock());
    registry().insert(d.get());
}

ThreadLocalBase::~ThreadLocalBase() {
    {
        std::lock_guard<std::mutex> g(registryLock());
        registry().erase(d.get());
    }
    for (auto &kv : d->values)
        d->destruct(kv.second);
}

void *ThreadLocalBase::get(bool &existed) {
    std::lock_guard<std::mutex> g(d->mutex);
    auto it = d->values.find(std::this_thread::get_id());
    if (it != d->values.end()) { existed = true; return it->second; }
    existed = false;
    void *v = d->construct();
    d->values[std::this_thread::get_id()] = v;
    return v;
}
const void *ThreadLocalBase::get(bool &existed) const { return const_cast<ThreadLocalBase *>(this)->get(existed); }
void *ThreadLocalBase::get() { bool e; return get(e); }
const void *ThreadLocalBase::get() const { bool e; return get(e); }

void initializeGlobalTLS() { }
void destroyGlobalTLS() { }
void initializeLocalTLS() { }
void destroyLocalTLS() {
    /* the calling thread is about to end: release its values */
    std::lock_guard<std::mutex> g(registryLock());
    for (Store *p : registry()) {
        void *v = nullptr;
        {
            std::lock_guard<std::mutex> g2(p->mutex);
            auto it = p->values.find(std::this_thread::get_id());
            if (it != p->values.end()) { v = it->second; p->values.erase(it); }
        }
        if (v) p->destruct(v);
    }
}

} // namespace detail
MTS_NAMESPACE_END
