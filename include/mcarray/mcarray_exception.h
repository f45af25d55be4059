// mca::MCArrayException -- same name and base as the reference (include/mcarray/mcarray_exception.h:51-56).
// Every non-zero status of the C ABI (include/mcarray_hip.h) is rethrown as this type.
#ifndef MCA_HIP_MCARRAY_EXCEPTION_H
#define MCA_HIP_MCARRAY_EXCEPTION_H
#include <stdexcept>
#include <string>
namespace mca {
class MCArrayException : public std::runtime_error {
public:
    explicit MCArrayException(const std::string &msg) : std::runtime_error(msg) {}
};
}  // namespace mca
#endif
