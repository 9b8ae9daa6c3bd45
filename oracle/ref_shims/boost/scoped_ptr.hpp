#pragma once
#include <memory>
namespace boost {
template <typename T> class scoped_ptr {
public:
    explicit scoped_ptr(T *p = nullptr) : m_p(p) { }
    scoped_ptr(const scoped_ptr &) = delete;
    scoped_ptr &operator=(const scoped_ptr &) = delete;
    void reset(T *p = nullptr) { m_p.reset(p); }
    T *get() const { return m_p.get(); }
    T *operator->() const { return m_p.get(); }
    T &operator*() const { return *m_p; }
    explicit operator bool() const { return (bool) m_p; }
    bool operator!() const { return !m_p; }
    void swap(scoped_ptr &o) { m_p.swap(o.m_p); }
private:
    std::unique_ptr<T> m_p;
};
}
