// tfhe/tfhe_garbage_collector.h -- client/gen_secure_keyset.cpp:3,19-21 registers parameter objects
// for deletion at exit; the shim's parameter objects live for the process lifetime, so this only
// has to accept them (and keep them reachable: leak checkers then stay quiet).
#ifndef REDSEC_TFHE_GC_SHIM_H
#define REDSEC_TFHE_GC_SHIM_H

struct LweParams;
struct TLweParams;
struct TGswParams;
struct TFheGateBootstrappingParameterSet;

#include <vector>

class TfheGarbageCollector {
 public:
  static void register_param(LweParams* p) { keep(p); }
  static void register_param(TLweParams* p) { keep(p); }
  static void register_param(TGswParams* p) { keep(p); }
  static void register_param(TFheGateBootstrappingParameterSet* p) { keep(p); }

 private:
  // registered objects stay reachable until the process ends (called from single-threaded set-up code, as in TFHE)
  static void keep(const void* p) {
    static std::vector<const void*>* const kept = new std::vector<const void*>();
    kept->push_back(p);
  }
};

#endif
