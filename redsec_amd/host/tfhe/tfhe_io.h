// tfhe/tfhe_io.h -- key and ciphertext files of the shim (call sites: client/*.cpp,
// nets/*/*/main.cpp:68-78, net.cpp:53-55).
//
// File formats are this backend's own (TFHE's on-disk format is a SURVEY.md section 8f "next"
// item): little-endian, a 16-byte header {magic "RSK1"/"RSS1"/ciphertext has none, ...} followed by
// raw arrays. A ciphertext record is int32 a[n], int32 b, double variance (4n + 12 bytes).
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
