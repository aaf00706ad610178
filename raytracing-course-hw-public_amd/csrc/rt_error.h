// rt_error.h — per-thread last-error string behind rt_last_error() (include/rt_abi.h). The reference reports
// failures by throwing std::runtime_error (caught at main.cpp:46-49); nothing is thrown across the C ABI.
#pragma once
#include <string>

namespace rt {
std::string &last_error();
int fail(int code, const std::string &msg);
} // namespace rt
