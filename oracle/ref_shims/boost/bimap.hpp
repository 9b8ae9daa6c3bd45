// ORACLE -- TEST INFRASTRUCTURE ONLY.  The sliver of Boost.Bimap that include/mitsuba/core/lrucache.h:64-146 uses (a set_of<K> left view, a
// list_of<int> right view in insertion / access order, one info value per relation), on std::map + std::list.  Written from the
// interface as lrucache.h calls it; Boost is not in this image.
#pragma once
#include <list>
#include <map>
#include <iterator>
namespace boost { namespace bimaps {
template <typename K, typename C = std::less<K> > struct set_of { typedef K key; typedef C compare; };
template <typename T> struct list_of { typedef T type; };
template <typename V> struct with_info { typedef V type; };

template <typename L, typename R, typename I> class bimap {
    typedef typename L::key K; typedef typename L::compare C; typedef typename R::type D; typedef typename I::type V;
public:
    struct relation { K first; D second; V info; relation(const K &k, const D &d, const V &v) : first(k), second(d), info(v) { } };
    typedef relation value_type;
private:
    typedef std::list<relation> list_type;
    typedef std::map<K, typename list_type::iterator, C> map_type;
    list_type m_list; map_type m_map;
public:
    struct left_iterator {
        typename map_type::iterator it;
        relation *operator->() const { return &*it->second; }
        relation &operator*() const { return *it->second; }
        bool operator==(const left_iterator &o) const { return it == o.it; }
        bool operator!=(const left_iterator &o) const { return it != o.it; }
    };
    typedef typename list_type::iterator right_iterator;
    typedef typename list_type::const_iterator right_const_iterator;
    typedef typename list_type::const_reverse_iterator right_const_reverse_iterator;
    struct left_view {
        bimap *b;
        left_iterator find(const K &k) { left_iterator i; i.it = b->m_map.find(k); return i; }
        left_iterator end() { left_iterator i; i.it = b->m_map.end(); return i; }
    } left;
    struct right_view {
        bimap *b;
        right_iterator begin() { return b->m_list.begin(); }
        right_iterator end() { return b->m_list.end(); }
        right_const_reverse_iterator rbegin() const { return b->m_list.rbegin(); }
        right_const_reverse_iterator rend() const { return b->m_list.rend(); }
        void relocate(right_iterator pos, right_iterator what) { b->m_list.splice(pos, b->m_list, what); }
        void erase(right_iterator what) { b->m_map.erase(what->first); b->m_list.erase(what); }
    } right;
    bimap() { left.b = this; right.b = this; }
    bimap(const bimap &o) : m_list(o.m_list) { left.b = this; right.b = this; for (right_iterator i = m_list.begin(); i != m_list.end(); ++i) m_map[i->first] = i; }
    bimap &operator=(const bimap &o) { m_list = o.m_list; m_map.clear(); for (right_iterator i = m_list.begin(); i != m_list.end(); ++i) m_map[i->first] = i; return *this; }
    size_t size() const { return m_list.size(); }
    right_iterator project_right(const left_iterator &i) { return i.it->second; }
    void insert(const relation &r) { if (m_map.find(r.first) != m_map.end()) return; m_list.push_back(r); m_map[r.first] = std::prev(m_list.end()); }
};
} }
