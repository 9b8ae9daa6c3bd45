#pragma once
#include <memory>
namespace boost {
template <typename T> class scoped_array {
public:
    explicit scoped_array(T *p = nullptr) : m_p(p) { }
    void reset(T *p = nullptr) { m_p.reset(p); }
    T *get() const { return m_p.get(); }
    T &operator[](std::ptrdiff_t i) const { return m_p[i]; }
private:
    std::unique_ptr<T[]> m_p;
};
}
