// tfhe/tfhe_io.h -- key and ciphertext files of the shim (call sites: client/*.cpp,
// nets/*/*/main.cpp:68-78, net.cpp:53-55).
//
// Ciphertext files use TFHE v1.1's own LweSample record: int32 type uid 42, int32 a[n], int32 b,
// double variance (4n + 16 bytes per sample), so image.ctxt / network_output.ctxt interchange with a
// TFHE client. KEY files are this backend's own format (little-endian, a 16-byte header with magic
// "RSK1"/"RSS1" followed by raw arrays): TFHE's key files (text property sections + typed binary
// arrays) cannot be checked against the real library here, see DESIGN.md section 7.
#ifndef REDSEC_TFHE_IO_SHIM_H
#define REDSEC_TFHE_IO_SHIM_H

#include <cstdio>

struct LweSample;
struct TFheGateBootstrappingParameterSet;
struct TFheGateBootstrappingCloudKeySet;
struct TFheGateBootstrappingSecretKeySet;

void export_tfheGateBootstrappingSecretKeySet_toFile(FILE* f, const TFheGateBootstrappingSecretKeySet* key);
void export_tfheGateBootstrappingCloudKeySet_toFile(FILE* f, const TFheGateBootstrappingCloudKeySet* key);
TFheGateBootstrappingSecretKeySet* new_tfheGateBootstrappingSecretKeySet_fromFile(FILE* f);
TFheGateBootstrappingCloudKeySet* new_tfheGateBootstrappingCloudKeySet_fromFile(FILE* f);
void export_gate_bootstrapping_ciphertext_toFile(FILE* f, const LweSample* sample, const TFheGateBootstrappingParameterSet* params);
void import_gate_bootstrapping_ciphertext_fromFile(FILE* f, LweSample* sample, const TFheGateBootstrappingParameterSet* params);

#endif
