// tfhe/tfhe_garbage_collector.h -- client/gen_secure_keyset.cpp:3,19-21 registers parameter objects
// for deletion at exit; the shim's parameter objects live for the process lifetime, so this only
// has to accept them.
#ifndef REDSEC_TFHE_GC_SHIM_H
#define REDSEC_TFHE_GC_SHIM_H

struct LweParams;
struct TLweParams;
struct TGswParams;
struct TFheGateBootstrappingParameterSet;

class TfheGarbageCollector {
 public:
  static void register_param(LweParams*) {}
  static void register_param(TLweParams*) {}
  static void register_param(TGswParams*) {}
  static void register_param(TFheGateBootstrappingParameterSet*) {}
};

#endif
