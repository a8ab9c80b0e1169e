// BinLayer.h -- binary-input layer (mirror of lib/BinLayer.h:35-75): same constructor, prep(),
// execute(), export_weights() and public in_dim/out_dim at the same offsets; the private part is an
// opaque pointer, so the object is never larger than the reference's and drivers compiled against
// either header link to this implementation.
#ifndef REDSEC_HOST_BINLAYER_H
#define REDSEC_HOST_BINLAYER_H

#include <cstdio>
#include "Layer.h"

namespace redsec_host { struct LayerImpl; }

class BinLayer {
 public:
  BinLayer(eConvType ec, uint16_t dep, ePoolType ep, eQuantType eq, tNetParams* np, TFheGateBootstrappingCloudKeySet* in_bk);
  tDimensions* prep(FILE* fd, tDimensions* dim);
  void* execute(tBit* p_in);
  void export_weights(FILE* fd_export);
  tDimensions in_dim;
  tDimensions out_dim;

 private:
  redsec_host::LayerImpl* impl;
};

#endif
