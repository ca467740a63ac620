"""FourCastNet v2-small (SFNO) on MI355X: host-side orchestration of the HIP building blocks of include/skyrim_sfno.h."""
