// gr::FDC::overlap_save — see fdc_blocks.h
#pragma once
#include "fdc_blocks.h"
