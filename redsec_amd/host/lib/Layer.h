// Layer.h -- shared types of the layer API (mirror of the reference's lib/Layer.h:31-188 for the
// ENCRYPTED flavour). Tag names, enumerator order and struct layouts are part of the link-level
// contract with the generated network drivers (nets/*/*/net.cpp), so they are restated exactly;
// everything behind the classes is this backend's own.
#ifndef REDSEC_HOST_LAYER_H
#define REDSEC_HOST_LAYER_H

#include <cstdint>
#include <tfhe/tfhe.h>
#include <tfhe/tfhe_io.h>

#define SIZE_EMPTY 1
#define SINGLE_BIT 1
#define MULTIBIT_BITS 12
#define FIXEDPOINT_BITS 12
#define MULTIBIT_SPACE 2048

typedef float tFloat;
typedef LweSample tBit;
typedef struct tMultiBits { tBit* ctxt; uint32_t size; } tMultiBit;
typedef tMultiBit tFixedPoint;

typedef enum _CONVTYPE { E_NO_CONV, E_CONV, E_FC, E_FC_FINAL, NUM_CONVS } eConvType;
typedef enum _POOLTYPE { E_NO_POOL, E_MAXPOOL, E_SUMPOOL, NUM_POOLS } ePoolType;
typedef enum _BIASTYPE { E_NO_BIAS, E_BIAS, E_BNORM, NUM_BIASES } eBiasType;
typedef enum _QUANT_TYPE { E_ACTIVATION_NONE, E_ACTIVATION_SIGN, E_ACTIVATION_RELU, NUM_ACTIVATIONS } eQuantType;
typedef enum _ACTION { E_INIT, E_PREP, E_EXEC, E_PREP_BIAS, E_EXPORT, NUM_ACTIONS } eAction;

typedef struct _WDSZ { int16_t h; int16_t w; } tRectangle;

typedef struct _DIMS {
  tRectangle hw;
  uint32_t in_dep;
  uint8_t in_bits, out_bits, filter_bits, bias_bits;
  uint32_t up_bound;
  float scale;
} tDimensions;   /* layout as lib/Layer.h:31-45: the reference's drivers are compiled against THEIR header */

typedef struct _CONV_PARAMS { tRectangle window; bool same_pad; float tern_thresh; tRectangle stride; } tConvParams;
typedef struct _BNORM_PARAMS { bool use_scale; float eps; } tBNormParams;
typedef struct _POOL_PARAMS { tRectangle window; bool same_pad; tRectangle stride; } tPoolParams;
typedef struct _QUANT_PARAMS { uint8_t shift_bits; } tQParams;
typedef struct _NET_PARAMS {
  tConvParams conv;
  tPoolParams pool;
  tBNormParams bnorm;
  tQParams quant;
  eBiasType e_bias;
  uint16_t version;
} tNetParams;
typedef union _ACT_PARAMS { tDimensions* d; tBit* b; tFixedPoint* fp; } tActParams;

uint64_t get_size(tRectangle* ws, uint16_t in_dep, uint16_t out_dep);
void netParamsCpy(tNetParams* dest, tNetParams* src);
tBit* bit_calloc(uint32_t len, TFheGateBootstrappingCloudKeySet* bk);
tMultiBit* mbit_calloc(uint32_t len, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk);
tFixedPoint* fixpt_calloc(uint32_t len, uint8_t bits, TFheGateBootstrappingCloudKeySet* bk);
void bit_free(uint32_t len, tBit* to_free);
void mbit_free(uint32_t len, tMultiBit* to_free);
void fixpt_free(uint32_t len, tFixedPoint* to_free);
void print_status(const char* s);

#endif
