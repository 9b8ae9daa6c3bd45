#pragma once
#include <mutex>
#include <chrono>
#include <condition_variable>
#include <thread>
#include <stdexcept>
#include <unistd.h>
namespace boost {
namespace posix_time {
    typedef std::chrono::steady_clock::time_point ptime;
    inline std::chrono::milliseconds milliseconds(long ms) { return std::chrono::milliseconds(ms); }
}
inline posix_time::ptime get_system_time() { return std::chrono::steady_clock::now(); }
template <typename M> struct scoped_lock_t : std::unique_lock<M> { using std::unique_lock<M>::unique_lock; };
struct mutex : std::mutex { typedef scoped_lock_t<std::mutex> scoped_lock; };
struct recursive_mutex : std::recursive_mutex { typedef scoped_lock_t<std::recursive_mutex> scoped_lock; };
struct timed_mutex : std::timed_mutex { typedef scoped_lock_t<std::timed_mutex> scoped_lock; };
struct recursive_timed_mutex : std::recursive_timed_mutex { typedef scoped_lock_t<std::recursive_timed_mutex> scoped_lock; };
template <typename M> using lock_guard = std::lock_guard<M>;
template <typename M> using unique_lock = std::unique_lock<M>;
struct condition_variable_any : std::condition_variable_any {
    template <typename L> bool timed_wait(L &lock, const posix_time::ptime &t) { return this->wait_until(lock, t) == std::cv_status::no_timeout; }
};
struct thread_interrupted { };
struct thread_resource_error : std::runtime_error { thread_resource_error() : std::runtime_error("thread_resource_error") { } };
}
