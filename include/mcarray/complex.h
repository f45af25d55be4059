// interleaved complex value types of the boundary (reference include/mcarray/complex.h:49-51)
#ifndef MCA_HIP_COMPLEX_H
#define MCA_HIP_COMPLEX_H
namespace mca {
typedef struct { float re; float im; } Complex32f;
typedef struct { double re; double im; } Complex64f;
typedef Complex64f Complex;
}  // namespace mca
#endif
