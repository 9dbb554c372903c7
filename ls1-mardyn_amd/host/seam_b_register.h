// seam_b_register.h — force-included (-include) when the UNMODIFIED reference Simulation.cpp is compiled for the
// device-resident build (oracle/ref_build/Makefile, target hipB).  The reference's own headers are included first, so
// that their later #include inside Simulation.cpp is a no-op (#pragma once / include guard) and the real classes stay
// declared under their own names; then the two construction sites
//     Simulation.cpp:177   _integrator = new Leapfrog();
//     Simulation.cpp:422   _moleculeContainer = new LinkedCells();
// are mapped to the device-resident classes of LinkedCellsHip.h.  The XML type strings ("LinkedCells", "Leapfrog") are
// string literals and are not affected.  No reference source file is edited or copied.
#pragma once
#include "integrators/Leapfrog.h"
#include "particleContainer/LinkedCells.h"

#include "LinkedCellsHip.h"

#define LinkedCells LinkedCellsHip
#define Leapfrog LeapfrogHip
