// IntFunc.h -- per-stage classes of the integer layers (mirror of lib/IntFunc.h:27-140, ENCRYPTED flavour); see BinFunc.h.
#ifndef REDSEC_HOST_INTFUNC_H
#define REDSEC_HOST_INTFUNC_H

#include <cstdio>
#include "Layer.h"

namespace redsec_host { struct LayerImpl; }

namespace IntFunc {

class Convolution {
 public:
  Convolution(uint16_t out_depth, tConvParams* in_params);
  tDimensions* prep(FILE* fd_filt, tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk);
  tFixedPoint* execute(tFixedPoint* p_inputs);

 private:
  redsec_host::LayerImpl* impl;
};

class SumPooling {
 public:
  SumPooling(tPoolParams* in_params);
  tDimensions* prep(tDimensions* ret_dim, TFheGateBootstrappingCloudKeySet* in_bk);
  tFixedPoint* execute(tFixedPoint* p_inputs);

 private:
  redsec_host::LayerImpl* impl;
};

class Quantize {
 public:
  Quantize(tQParams* qparam);
  tDimensions* prep(FILE* fd_bias, tDimensions* ret_dim, tMultiBit* p_bias, uint32_t* p_slope, TFheGateBootstrappingCloudKeySet* in_bk);
  tBit* execute(tFixedPoint* p_inputs, tMultiBit* p_bias);
  tFixedPoint* add_bias(tFixedPoint* p_inputs, tMultiBit* p_bias);
  tFixedPoint* relu_shift(tFixedPoint* p_inputs, tMultiBit* p_bias, uint32_t* p_slope);

 private:
  redsec_host::LayerImpl* impl;
};

}  // namespace IntFunc

#endif
