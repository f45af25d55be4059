// Boundary value types (reference include/mcarray/mcadefs.h:56-89).  The reference's SignalPtr is
// boost::shared_array<double>; boost is not a dependency here, so mca::shared_array<T> provides the
// members the reference and its callers use (get(), operator[], reset(p), reset(p, deleter), use_count()).
// Define MCA_USE_BOOST to get the reference's exact typedefs when boost is available.
#ifndef MCA_HIP_MCADEFS_H
#define MCA_HIP_MCADEFS_H
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "complex.h"

#ifdef MCA_USE_BOOST
#include <boost/shared_array.hpp>
#endif

namespace mca {

#ifdef MCA_USE_BOOST
template <typename T> using shared_array = boost::shared_array<T>;
#else
template <typename T>
class shared_array {
public:
    shared_array() {}
    explicit shared_array(T *p) : p_(p, std::default_delete<T[]>()) {}
    template <typename D> shared_array(T *p, D d) : p_(p, d) {}
    void reset() { p_.reset(); }
    void reset(T *p) { p_.reset(p, std::default_delete<T[]>()); }
    template <typename D> void reset(T *p, D d) { p_.reset(p, d); }
    T *get() const { return p_.get(); }
    T &operator[](std::ptrdiff_t i) const { return p_.get()[i]; }
    long use_count() const { return p_.use_count(); }
    explicit operator bool() const { return static_cast<bool>(p_); }
private:
    std::shared_ptr<T> p_;
};
#endif

typedef float BaseType32;
typedef Complex32f BaseType32C;
typedef shared_array<BaseType32> SignalPtr32;
typedef std::vector<SignalPtr32> SignalVector32;
typedef double BaseType64;
typedef Complex64f BaseType64C;
typedef signed short BaseType16s;
typedef shared_array<BaseType16s> SignalPtr16s;
typedef std::vector<SignalPtr16s> SignalVector16s;
typedef BaseType64 BaseType;
typedef BaseType64C BaseTypeC;
typedef shared_array<BaseType> SignalPtr;
typedef shared_array<BaseTypeC> SignalCPtr;
typedef std::vector<SignalPtr> SignalVector;
typedef std::vector<SignalCPtr> SignalCVector;

}  // namespace mca
#endif
