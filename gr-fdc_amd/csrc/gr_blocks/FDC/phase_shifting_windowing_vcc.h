// gr::FDC::phase_shifting_windowing_vcc — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
