// ref_logits_driver.cpp -- TEST INFRASTRUCTURE. A main() of OUR OWN for the reference's plaintext flavour.
//
// The reference's own nets/*/*/main.cpp prints the logits for sign1024x1 only (main.cpp:132) and runs the
// ReLU nets on ONE image (NUM_SAMPLES, nets/mnist/relu1024x1/main.cpp:25). This driver links the
// reference's UNMODIFIED lib/*.cpp (plaintext flavour) and a net's UNMODIFIED net.cpp -- compiled where they
// lie under /root/reference by oracle/Makefile, objects into oracle/_ref/ only -- constructs its HeBNN, feeds
// the first rows of a csv file through HeBNN::run and prints every logit, so that tests/golden/ can hold
// reference outputs for every image and every net (tests/golden/make_golden.py).
//
//   usage (cwd = the net's directory, where HeBNN() opens var_prep.dat):
//     <net>_logits.out <csv> <rows> sign|relu
//   preprocessing:  sign  2 v - 255        (nets/mnist/sign1024x1/main.cpp:155, client/encrypt_image.cpp:76)
//                   relu  v / 100 - 1      (nets/mnist/relu1024x1/main.cpp:203)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "net.h"   // the net's own header, found through -I<reference net dir>

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <csv> <rows> sign|relu\n", argv[0]); return 2; }
  const int rows = atoi(argv[2]);
  const bool relu = strcmp(argv[3], "relu") == 0;
  HeBNN* net = new HeBNN();
  tDimensions in, out;
  net->get_in_dims(&in);
  net->get_out_dims(&out);
  const size_t len = (size_t)in.hw.h * in.hw.w * in.in_dep;
  FILE* f = fopen(argv[1], "r");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  std::vector<char> line(16 * len + 64);
  for (int r = 0; r < rows && fgets(line.data(), (int)line.size(), f); ++r) {
    if (line[0] < '0' || line[0] > '9') { --r; continue; }
    tFixedPoint* x = (tFixedPoint*)calloc(len, sizeof(tFixedPoint));   // HeBNN::run frees its input
    char* tok = strtok(line.data(), ",");
    const int label = atoi(tok);
    for (size_t i = 0; i < len; ++i) {
      tok = strtok(NULL, ",\n");
      if (tok && *tok) x[i] = relu ? (tFixedPoint)(atoi(tok) / 100 - 1) : (tFixedPoint)(2 * atoi(tok) - 255);
    }
    tFixedPoint* y = (tFixedPoint*)net->run(x);
    printf("row %d label %d logits", r, label);
    for (int k = 0; k < (int)out.in_dep; ++k) printf(" %d", (int)y[k]);
    printf("\n");
    free(y);
  }
  fclose(f);
  return 0;
}
