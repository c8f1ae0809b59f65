#!/bin/bash
cd tools
for n in 262144 2097152; do
for gib in 0.5 4 16; do
for mode in 1 4; do ./randline $gib $n $mode; done
./randline $gib $n 2 262144
./randline $gib $n 2 1
done; done
./randline 16 262144 1 0 1
./randline 16 262144 2 262144 1
