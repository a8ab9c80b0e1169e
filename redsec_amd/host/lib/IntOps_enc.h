// IntOps_enc.h -- per-ciphertext integer primitives (mirror of lib/IntOps_enc.h:9-32).
#ifndef REDSEC_HOST_INTOPS_ENC_H
#define REDSEC_HOST_INTOPS_ENC_H

#include "Layer.h"

namespace IntOps {
void invert(tFixedPoint* result, const tFixedPoint* a, const uint8_t* b, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk);
void add(tFixedPoint* result, const tFixedPoint* a, const tFixedPoint* b, uint8_t in1_bits, TFheGateBootstrappingCloudKeySet* bk);
void add_inplace(tFixedPoint* result, const tFixedPoint* a, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk);
void subtract(tFixedPoint* result, const tFixedPoint* a, const tFixedPoint* b, uint8_t in1_bits, TFheGateBootstrappingCloudKeySet* bk);
void relu(tFixedPoint* result, tFixedPoint* in1, uint8_t input_bits, TFheGateBootstrappingCloudKeySet* bk);
void shift(tFixedPoint* result, tFixedPoint* in1, uint8_t input_bits, uint8_t shift_bits, TFheGateBootstrappingCloudKeySet* bk);
}  // namespace IntOps

#endif
