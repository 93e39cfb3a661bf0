// The row-split form of the hidden-256 path's kernel A (objnerf_train256r_body.h) as a translation unit of its own: the
// shared part of objnerf_train256.hip (types, tables, arguments, helpers) + that kernel, compiled WITHOUT
// -mllvm -amdgpu-mfma-vgpr-form (see the note in objnerf_train256.hip).  It is the DEFAULT form of kernel A (OBJ256_FIRST_FORM=1, a diagnostic switch, selects fwd256_kernel instead).
#define OBJ256_ROWSPLIT_TU
#include "objnerf_train256.hip"
