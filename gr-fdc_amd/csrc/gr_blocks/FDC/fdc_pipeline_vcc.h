// gr::FDC::fdc_pipeline_vcc — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
