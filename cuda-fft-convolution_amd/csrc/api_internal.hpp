// api_internal.hpp -- shared by the host translation units of libfftconv.so (not installed).
#pragma once
#include <string>

namespace fc {
// records the calling thread's last-error message (fftconv_last_error) and returns `code`
int api_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// the calling thread's last-error message / overwrite it (to carry a worker thread's error to the caller)
std::string api_last_error();
void api_set_last_error(const std::string& msg);
}  // namespace fc
