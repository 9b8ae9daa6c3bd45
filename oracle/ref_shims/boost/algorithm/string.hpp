#pragma once
#include <string>
#include <algorithm>
#include <cctype>
namespace boost {
inline bool starts_with(const std::string &s, const std::string &p) { return s.size() >= p.size() && s.compare(0, p.size(), p) == 0; }
inline bool ends_with(const std::string &s, const std::string &p) { return s.size() >= p.size() && s.compare(s.size() - p.size(), p.size(), p) == 0; }
inline std::string to_lower_copy(std::string s) { std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char) std::tolower(c); }); return s; }
inline std::string to_upper_copy(std::string s) { std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char) std::toupper(c); }); return s; }
inline void to_lower(std::string &s) { s = to_lower_copy(s); }
inline void to_upper(std::string &s) { s = to_upper_copy(s); }
inline std::string trim_copy(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char) s[a])) ++a;
    while (b > a && std::isspace((unsigned char) s[b - 1])) --b;
    return s.substr(a, b - a);
}
inline void trim(std::string &s) { s = trim_copy(s); }
namespace algorithm { using boost::to_lower_copy; using boost::to_upper_copy; using boost::starts_with; using boost::ends_with; }
}
