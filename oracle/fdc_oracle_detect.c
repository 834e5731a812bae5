/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Detector / activation restatements (filled in below). */
#include "fdc_oracle.h"
