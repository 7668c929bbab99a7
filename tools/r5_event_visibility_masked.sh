#!/bin/bash
# the two-process probe again, (a) both processes on all CUs, (b) on DISJOINT halves of the chip (HSA_CU_MASK): does the corruption of
# the streaming kernel's output need the two processes to SHARE compute units?
N=${N:-2000}
echo "-- (a) both on all CUs"
python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P1=$!
python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P2=$!
wait $P1 $P2
echo "-- (b) disjoint CU halves"
HSA_CU_MASK="0:0-127" python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P1=$!
HSA_CU_MASK="0:128-255" python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P2=$!
wait $P1 $P2
echo "-- (c) both on all CUs again"
python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P1=$!
python tools/r5_event_visibility.py $N conv event 2>/dev/null | grep "stale$" & P2=$!
wait $P1 $P2
