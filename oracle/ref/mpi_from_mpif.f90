! TEST INFRASTRUCTURE (oracle/_ref build only; never linked into the product).
!
! The reference sources say `use mpi`. The image ships a real MPICH 3.3.2 under
! /opt/conda (libmpi.so.12, libmpifort.so.12, include/mpif.h), but its `mpi.mod`
! was written by gfortran and cannot be read by amdflang. This file generates the
! module from MPICH's OWN Fortran header, so the reference is compiled and linked
! against the real MPI library (no stand-in symbols are written here).
module mpi
  implicit none
  include 'mpif.h'
end module mpi
