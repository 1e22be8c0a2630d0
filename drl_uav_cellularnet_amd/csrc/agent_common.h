// Shared by the two translation units of libuavagent.so (not part of the C ABI: hidden visibility).
#pragma once
#include <string>

namespace uavagent_internal {
// Records the thread-local message uavagent_last_error() returns and hands `code` back.
__attribute__((visibility("hidden"))) int fail(int code, const std::string &msg);
}  // namespace uavagent_internal
