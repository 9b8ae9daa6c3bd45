#pragma once
#include <filesystem>
#include <fstream>
#include <unistd.h>
#include <cassert>
namespace boost { namespace filesystem {
/* the reference forward-declares `class boost::filesystem::path` (core/fwd.h:181), so it has to be a class of that name */
class path : public std::filesystem::path {
public:
    typedef std::filesystem::path base;
    path() { }
    path(const base &p) : base(p) { }
    path(base &&p) : base(std::move(p)) { }
    path(const std::string &s) : base(s) { }
    path(const char *s) : base(s) { }
    path parent_path() const { return path(base::parent_path()); }
    path filename() const { return path(base::filename()); }
    path stem() const { return path(base::stem()); }
    path extension() const { return path(base::extension()); }
    path &operator/=(const path &o) { base::operator/=(o); return *this; }
    path &replace_extension(const path &e = path()) { base::replace_extension(e); return *this; }
};
inline path operator/(const path &a, const path &b) { return path(static_cast<const path::base &>(a) / static_cast<const path::base &>(b)); }
inline bool exists(const path &p) { return std::filesystem::exists(p); }
inline bool is_directory(const path &p) { return std::filesystem::is_directory(p); }
inline bool is_regular_file(const path &p) { return std::filesystem::is_regular_file(p); }
inline bool remove(const path &p) { return std::filesystem::remove(p); }
inline std::uintmax_t file_size(const path &p) { return std::filesystem::file_size(p); }
inline void resize_file(const path &p, std::uintmax_t n) { std::filesystem::resize_file(p, n); }
inline path absolute(const path &p) { return path(std::filesystem::absolute(p)); }
inline path canonical(const path &p) { return path(std::filesystem::canonical(p)); }
inline path current_path() { return path(std::filesystem::current_path()); }
inline bool create_directory(const path &p) { return std::filesystem::create_directory(p); }
using std::filesystem::directory_iterator;
class ifstream : public std::ifstream { public: ifstream() { } explicit ifstream(const path &p, std::ios_base::openmode m = std::ios_base::in) : std::ifstream(p.string(), m) { } void open(const path &p, std::ios_base::openmode m = std::ios_base::in) { std::ifstream::open(p.string(), m); } };
class ofstream : public std::ofstream { public: ofstream() { } explicit ofstream(const path &p, std::ios_base::openmode m = std::ios_base::out) : std::ofstream(p.string(), m) { } void open(const path &p, std::ios_base::openmode m = std::ios_base::out) { std::ofstream::open(p.string(), m); } };
} }
#include <system_error>
#include <chrono>
namespace boost { namespace system { typedef std::error_code error_code; } }
namespace boost { namespace filesystem {
inline std::time_t last_write_time(const path &p, boost::system::error_code &ec) {
    auto t = std::filesystem::last_write_time(p, ec);
    return (std::time_t) std::chrono::duration_cast<std::chrono::seconds>(t.time_since_epoch()).count();
}
} }
