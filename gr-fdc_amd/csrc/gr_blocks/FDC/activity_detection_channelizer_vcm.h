// gr::FDC::activity_detection_channelizer_vcm — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
