// IntLayer.h -- integer-input layer (mirror of lib/IntLayer.h:35-80); see BinLayer.h.
#ifndef REDSEC_HOST_INTLAYER_H
#define REDSEC_HOST_INTLAYER_H

#include <cstdio>
#include "Layer.h"

namespace redsec_host { struct LayerImpl; }

class IntLayer {
 public:
  IntLayer(eConvType ec, uint16_t dep, ePoolType ep, eQuantType eq, tNetParams* np, TFheGateBootstrappingCloudKeySet* in_bk);
  tDimensions* prep(FILE* fd, tDimensions* dim);
  void* execute(tMultiBit* p_in);
  void export_weights(FILE* fd);
  tDimensions in_dim;
  tDimensions out_dim;

 private:
  redsec_host::LayerImpl* impl;
};

#endif
