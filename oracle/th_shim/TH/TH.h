/* Minimal stand-in for the Torch7 TH header, just enough to compile the
 * reference's lib/layer_utils/roi_pooling/src/roi_pooling.c unmodified
 * (oracle/Makefile target `ref`).  Test infrastructure only. */
#ifndef SIS3D_TH_SHIM_H
#define SIS3D_TH_SHIM_H
#include <float.h>
typedef struct THFloatTensor { float *data; long size[8]; } THFloatTensor;
static inline float *THFloatTensor_data(THFloatTensor *t) { return t->data; }
static inline long THFloatTensor_size(THFloatTensor *t, int d) { return t->size[d]; }
#endif
