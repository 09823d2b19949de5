// engine/env.hpp -- The engine's environment switches, read through ONE door so that a run can say which were set.
// Every switch selects between code paths that give the same bits; none is needed for normal use.
// But a stray DES_PATCH=0 changes what a benchmark measures, so des_env::get() remembers every switch it found SET and
// des_dev_config_string() (include/des_dev.h) hands the list to the caller: bench.py prints it in its JSON line.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace des_env {

inline std::mutex &mu() { static std::mutex m; return m; }
inline std::map<std::string, std::string> &seen() { static std::map<std::string, std::string> s; return s; }

// std::getenv(name); a variable that is set is recorded (name -> value as read)
inline const char *get(const char *name)
{
    const char *v = std::getenv(name);
    if (v) { std::lock_guard<std::mutex> g(mu()); seen()[name] = v; }
    return v;
}

// "NAME=value NAME=value ..." in name order; empty when no switch was set
inline std::string summary()
{
    std::lock_guard<std::mutex> g(mu());
    std::string out;
    for (const auto &kv : seen()) { if (!out.empty()) out += ' '; out += kv.first + "=" + kv.second; }
    return out;
}

} // namespace des_env
