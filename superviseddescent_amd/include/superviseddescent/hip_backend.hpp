// superviseddescent/hip_backend.hpp -- RAII handle over the C-ABI (include/sdm.h) for the header layer.
//
// Status codes of the C-ABI are turned back into the exceptions the reference's headers throw
// (std::runtime_error: include/rcr/helpers.hpp:143-145, include/rcr/model.hpp:197-200), so code written
// against the reference keeps its error handling.  There is no CPU fallback behind this handle: without
// libsdm_hip.so's device the constructor throws.
#pragma once

#ifndef SDM_HIP_BACKEND_HPP_
#define SDM_HIP_BACKEND_HPP_

#include "sdm.h"

#include <memory>
#include <stdexcept>
#include <string>

namespace superviseddescent {
namespace hip {

inline void check(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + sdm_last_error());
}

class Handle {
public:
    explicit Handle(int device = 0) : ctx_(sdm_create(device))
    {
        if (!ctx_) throw std::runtime_error(std::string("sdm_create: ") + sdm_last_error());
    }
    ~Handle() { if (ctx_) sdm_destroy(ctx_); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    sdm_ctx* get() const { return ctx_; }

private:
    sdm_ctx* ctx_;
};

// one lazily created handle per thread for the stand-alone solver calls
inline Handle& default_handle()
{
    static thread_local std::unique_ptr<Handle> h;
    if (!h) h.reset(new Handle(0));
    return *h;
}

}  // namespace hip
}  // namespace superviseddescent

#endif
