// seam_b_register.h — force-included (-include) when the UNMODIFIED reference Simulation.cpp is compiled for the
// device-resident build (oracle/ref_build/Makefile, target hipB).  The reference's own headers are included first, so
// that their later #include inside Simulation.cpp is a no-op (#pragma once / include guard) and the real classes stay
// declared under their own names; then the two construction sites
//     Simulation.cpp:177   _integrator = new Leapfrog();
//     Simulation.cpp:422   _moleculeContainer = new LinkedCells();
// are mapped to the device-resident classes of LinkedCellsHip.h.  The XML type strings ("LinkedCells", "Leapfrog") are
// string literals and are not affected.  No reference source file is edited or copied.
// Round 3, multi-rank seam: the third construction site
//     Simulation.cpp:1356  _domainDecomposition = new DomainDecompBase();     (non-MPI build)
// becomes `new DomainDecompHip()` through a FUNCTION-LIKE macro: it only fires where the name is followed by "(", i.e. at
// that constructor call — declarations such as `DomainDecompBase* domainDecomposition` (Simulation.cpp:1321) keep the base
// type.  The headers that declare the class are included first, under their real names.
#pragma once
#include "Simulation.h"
#include "integrators/Leapfrog.h"
#include "parallel/DomainDecompBase.h"
#include "particleContainer/LinkedCells.h"

#include "DomainDecompHip.h"
#include "LinkedCellsHip.h"

#define LinkedCells LinkedCellsHip
#define Leapfrog LeapfrogHip
#define DomainDecompBase(...) DomainDecompHip(__VA_ARGS__)
