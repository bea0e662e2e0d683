// data.h of the reference: the scene-description loader (data.h:15).
#pragma once

// Reads the .dat file `name`, builds its variables in memory and returns the one the file
// exports (caller-owned, never freed).  Errors go to HW_error, which terminates.
void *D_data(char *name);
