// guard.h -- the exception firewall of the extern "C" boundary: internal failures become a status code plus a thread-local
// message (eesen_last_error); nothing throws across the C-ABI.
#pragma once
#include "common.h"

namespace eesen {

std::string& last_error_slot();  // thread-local, defined in capi.cpp

template <class F>
int guard(F f) {
  try {
    f();
    return EESEN_OK;
  } catch (const Error& e) {
    last_error_slot() = e.what();
    return e.code;
  } catch (const std::exception& e) {
    last_error_slot() = e.what();
    return EESEN_ERR_INVALID;
  } catch (...) {
    last_error_slot() = "unknown failure";
    return EESEN_ERR_INVALID;
  }
}
#define REQ_PTR(p) EESEN_REQUIRE((p) != nullptr, EESEN_ERR_INVALID, "null pointer argument")

}  // namespace eesen
