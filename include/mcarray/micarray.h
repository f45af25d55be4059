// umbrella header (reference include/mcarray/micarray.h)
#ifndef MCA_HIP_MICARRAY_H
#define MCA_HIP_MICARRAY_H
#include "ArrayDescription.h"
#include "ArrayModules.h"
#include "BinauralLocalisation.h"
#include "FastBinauralMasking.h"
#include "MultibandBinarualLocalisation.h"
#include "MvdrBeamformer.h"
#include "Beamformer.h"
#include "BeamformingSeparationAndLocalistaion.h"
#include "SoundLocalisationCallback.h"
#include "SourceLocalisation.h"
#include "SourceSeparationAndLocalisation.h"
#include "SteeringBeamforming.h"
#include "mcadefs.h"
#include "mcarray_exception.h"
#include "microhponeArrayHelpers.h"
#endif
