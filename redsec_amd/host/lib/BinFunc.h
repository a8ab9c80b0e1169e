// BinFunc.h -- per-stage classes of the binary layers (mirror of lib/BinFunc.h:37-173, ENCRYPTED flavour): the same
// class names, constructors and prep / execute / add_bias / relu_shift signatures; the private parts are one opaque
// pointer. Each execute() is ONE batched launch sequence on the device (the reference runs an OpenMP loop of
// per-ciphertext TFHE calls, e.g. lib/BinFunc.cpp:1056-1071), frees its input as the reference's callee does and keeps its
// output resident for the next stage. BatchNorm and the extract_* / export_* members exist only in the reference's
// weight-convert flavour (out of scope).
#ifndef REDSEC_HOST_BINFUNC_H
#define REDSEC_HOST_BINFUNC_H

#include <cstdio>
#include "Layer.h"

namespace redsec_host { struct LayerImpl; }

namespace BinFunc {

class Convolution {
 public:
  Convolution(uint32_t out_depth, tConvParams* in_params);
  tDimensions* prep(FILE* fd_filt, tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk);
  tMultiBit* execute(tBit* p_inputs);
  void get_outhw(tRectangle* ret_dim);
  void get_outdep(uint32_t* ret_dep);

 private:
  redsec_host::LayerImpl* impl;
};

class SumPooling {
 public:
  SumPooling(tPoolParams* in_params);
  tDimensions* prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk);
  tMultiBit* execute(tMultiBit* p_inputs);
  void get_outhw(tRectangle* ret_dim);
  void get_outdep(uint32_t* ret_dep);

 private:
  redsec_host::LayerImpl* impl;
};

class MaxPooling {
 public:
  MaxPooling(tPoolParams* in_params);
  tDimensions* prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk);
  tBit* execute(tBit* p_inputs);

 private:
  redsec_host::LayerImpl* impl;
};

class Quantize {
 public:
  Quantize(tQParams* qparam);
  tDimensions* prep(FILE* fd_bias, tDimensions* ret_dim, tMultiBit* p_bias, uint32_t* p_slope, TFheGateBootstrappingCloudKeySet* in_bk);
  tBit* execute(tMultiBit* p_inputs, tMultiBit* p_bias);
  tMultiBit* add_bias(tMultiBit* p_inputs, tMultiBit* p_bias);
  tFixedPoint* relu_shift(tMultiBit* p_inputs, tMultiBit* p_bias, uint32_t* p_slope);

 private:
  redsec_host::LayerImpl* impl;
};

}  // namespace BinFunc

#endif
