// types.hpp -- the scalar typedefs of the boundary (reference libs/types.hpp:4-6).
#ifndef SVC_TYPES_HPP
#define SVC_TYPES_HPP

typedef unsigned int uint;
typedef unsigned char uchar;

#endif  // SVC_TYPES_HPP
