# Install/unInstall the USER-EDM package files in LAMMPS (run by "make yes-user-edm" /
# "make no-user-edm" from the LAMMPS src directory).  The fixes link against the locally
# installed EDM library: add -I<prefix>/include and -L<prefix>/lib -ledm -ledm_hip to the
# machine makefile.

if (test $1 = 1) then

  cp fix_edm.cpp ..
  cp fix_edm.h ..
  cp fix_edm_pair.cpp ..
  cp fix_edm_pair.h ..

elif (test $1 = 0) then

  rm -f ../fix_edm.cpp
  rm -f ../fix_edm.h
  rm -f ../fix_edm_pair.cpp
  rm -f ../fix_edm_pair.h

fi
