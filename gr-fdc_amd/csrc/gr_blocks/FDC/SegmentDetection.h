// gr::FDC::SegmentDetection — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
