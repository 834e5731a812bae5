// gr::FDC::vector_cut_vxx — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
