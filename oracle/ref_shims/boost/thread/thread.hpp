#pragma once
#include <boost/thread/mutex.hpp>
#include <system_error>
namespace boost {
class thread {
public:
    thread() { }
    template <typename F, typename... A> explicit thread(F &&f, A &&...a) {
        try { m_t = std::thread(std::forward<F>(f), std::forward<A>(a)...); } catch (const std::system_error &) { throw thread_resource_error(); }
    }
    thread(thread &&o) = default;
    thread &operator=(thread &&o) { if (m_t.joinable()) m_t.detach(); m_t = std::move(o.m_t); return *this; }
    ~thread() { if (m_t.joinable()) m_t.detach(); }
    void join() { if (m_t.joinable()) m_t.join(); }
    void detach() { if (m_t.joinable()) m_t.detach(); }
    bool joinable() const { return m_t.joinable(); }
    std::thread::native_handle_type native_handle() { return m_t.native_handle(); }
private:
    std::thread m_t;
};
namespace this_thread {
    inline void yield() { std::this_thread::yield(); }
    template <typename D> void sleep(const D &d) { std::this_thread::sleep_for(d); }
}
}
