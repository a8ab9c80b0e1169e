// keyio_roundtrip.cpp -- test program (tests/test_tfhe_keyio.py): reads a secret key file and a cloud key file
// through the shim's TFHE-API readers (new_tfheGateBootstrapping{Secret,Cloud}KeySet_fromFile: format detected
// from the first byte) and writes both back through the writers (format from REDSEC_KEY_FORMAT), the way the
// reference's tools use them (client/gen_secure_keyset.cpp:108-114, nets/mnist/sign1024x1/net.cpp:53-55).
// Never touches the GPU.
#include <tfhe/tfhe.h>
#include <tfhe/tfhe_io.h>

int main(int argc, char** argv) {
  if (argc != 5) { fprintf(stderr, "usage: %s secret.in cloud.in secret.out cloud.out\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  TFheGateBootstrappingSecretKeySet* sk = new_tfheGateBootstrappingSecretKeySet_fromFile(f);
  fclose(f);
  f = fopen(argv[2], "rb");
  if (!f) return 3;
  TFheGateBootstrappingCloudKeySet* ck = new_tfheGateBootstrappingCloudKeySet_fromFile(f);
  fclose(f);
  printf("n=%d N=%d l=%d Bgbit=%d t=%d basebit=%d\n", sk->params->in_out_params->n, sk->params->tgsw_params->tlwe_params->N,
         sk->params->tgsw_params->l, sk->params->tgsw_params->Bgbit, ck->params->ks_t, ck->params->ks_basebit);
  f = fopen(argv[3], "wb");
  export_tfheGateBootstrappingSecretKeySet_toFile(f, sk);
  fclose(f);
  f = fopen(argv[4], "wb");
  export_tfheGateBootstrappingCloudKeySet_toFile(f, ck);
  fclose(f);
  delete_gate_bootstrapping_secret_keyset(sk);
  delete_gate_bootstrapping_cloud_keyset(ck);
  return 0;
}
